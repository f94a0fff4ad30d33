// Host side of libttx_hip.so: the C ABI of include/ttx.h, the runtime around the kernels (sessions, workspaces, hipGraphs,
// polling loops, slot pools) and the loop / bookkeeping kernels (ttx_loop_kernels.hip.h).  The GEMM and attention families
// live in ttx_gemm.hip / ttx_attn.hip and are reached through the launchers of ttx_internal.h.
// No CPU fallback exists anywhere in this library: without a gfx950 device every entry point fails.
#include "ttx_internal.h"
#include "ttx_loop_kernels.hip.h"
#include "ttx_tokenizer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace ttx;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int ttx::fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

static void layout_add(ttx_model* m, const std::string& name, size_t numel, size_t* off_out) {
  size_t off = (m->blob_floats + 63) & ~(size_t)63;
  m->index[name] = {off, numel};
  m->blob_floats = off + numel;
  if (off_out) *off_out = off;
}

static int build_layout(ttx_model* m) {
  const ttx_config& c = m->cfg;
  const size_t d = c.embedding_dim, F = c.feedforward_dim, V = c.vocab_size, Vs = c.src_vocab_size;
  if (c.embedding_dim <= 0 || c.embedding_dim % 64 || c.embedding_dim > 1024)
    return fail(TTX_ERR_INVALID, "embedding_dim must be a multiple of 64 in [64,1024]");
  if (c.num_heads <= 0 || c.embedding_dim / c.num_heads != ATT_DH || c.embedding_dim % c.num_heads)
    return fail(TTX_ERR_INVALID, "embedding_dim / num_heads must be 32");
  const int vpl = c.embedding_dim / 64;
  if (vpl != 1 && vpl != 2 && vpl != 4 && vpl != 8 && vpl != 16)
    return fail(TTX_ERR_INVALID, "embedding_dim must be 64, 128, 256, 512 or 1024");
  if (c.feedforward_dim <= 0 || c.feedforward_dim % 64) return fail(TTX_ERR_INVALID, "feedforward_dim must be a multiple of 64");
  if (c.vocab_size <= 0 || c.src_vocab_size <= 0) return fail(TTX_ERR_INVALID, "vocab sizes must be positive");
  if (c.num_encoder_layers <= 0 || c.num_decoder_layers <= 0) return fail(TTX_ERR_INVALID, "layer counts must be positive");
  if (c.max_positions <= 0) return fail(TTX_ERR_INVALID, "max_positions must be positive");

  auto attn = [&](const std::string& p, size_t* iw, size_t* ib, size_t* ow, size_t* ob) {
    layout_add(m, p + ".in_proj_weight", 3 * d * d, iw);
    layout_add(m, p + ".in_proj_bias", 3 * d, ib);
    layout_add(m, p + ".out_proj.weight", d * d, ow);
    layout_add(m, p + ".out_proj.bias", d, ob);
  };
  auto ffn = [&](const std::string& p, LayerW& w) {
    layout_add(m, p + ".linear1.weight", F * d, &w.l1_w);
    layout_add(m, p + ".linear1.bias", F, &w.l1_b);
    layout_add(m, p + ".linear2.weight", d * F, &w.l2_w);
    layout_add(m, p + ".linear2.bias", d, &w.l2_b);
  };
  auto norm = [&](const std::string& p, size_t* w, size_t* b) {
    layout_add(m, p + ".weight", d, w);
    layout_add(m, p + ".bias", d, b);
  };
  layout_add(m, "src_token_featurizer.embedding.weight", Vs * d, &m->src_emb);
  layout_add(m, "tgt_token_featurizer.embedding.weight", V * d, &m->tgt_emb);
  m->enc.resize(c.num_encoder_layers);
  for (int i = 0; i < c.num_encoder_layers; ++i) {
    const std::string p = "transformer.encoder.layers." + std::to_string(i);
    LayerW& w = m->enc[i];
    attn(p + ".self_attn", &w.sa_in_w, &w.sa_in_b, &w.sa_out_w, &w.sa_out_b);
    ffn(p, w);
    norm(p + ".norm1", &w.n1_w, &w.n1_b);
    norm(p + ".norm2", &w.n2_w, &w.n2_b);
  }
  norm("transformer.encoder.norm", &m->enc_norm_w, &m->enc_norm_b);
  m->dec.resize(c.num_decoder_layers);
  for (int i = 0; i < c.num_decoder_layers; ++i) {
    const std::string p = "transformer.decoder.layers." + std::to_string(i);
    LayerW& w = m->dec[i];
    attn(p + ".self_attn", &w.sa_in_w, &w.sa_in_b, &w.sa_out_w, &w.sa_out_b);
    attn(p + ".multihead_attn", &w.ca_in_w, &w.ca_in_b, &w.ca_out_w, &w.ca_out_b);
    ffn(p, w);
    norm(p + ".norm1", &w.n1_w, &w.n1_b);
    norm(p + ".norm2", &w.n2_w, &w.n2_b);
    norm(p + ".norm3", &w.n3_w, &w.n3_b);
  }
  norm("transformer.decoder.norm", &m->dec_norm_w, &m->dec_norm_b);
  layout_add(m, "next_token_classifier.weight", V * d, &m->cls_w);
  layout_add(m, "next_token_classifier.bias", V, &m->cls_b);
  // derived tensors (not in the state dict)
  layout_add(m, "positional_encoding.pe", ((size_t)c.max_positions + 1) * d, &m->pe);
  layout_add(m, "derived.cross_kv.weight", (size_t)c.num_decoder_layers * 2 * d * d, &m->cross_kv_w);
  layout_add(m, "derived.cross_kv.bias", (size_t)c.num_decoder_layers * 2 * d, &m->cross_kv_b);
  m->blob_floats = (m->blob_floats + 63) & ~(size_t)63;
  return TTX_OK;
}

static bool is_gfx950(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" int ttx_abi_version(void) { return TTX_ABI_VERSION; }
extern "C" const char* ttx_last_error(void) { return g_err.c_str(); }

extern "C" int ttx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) ok += is_gfx950(i) ? 1 : 0;
  return ok;
}

static int model_alloc(const ttx_config* cfg, int device, ttx_model** out) {
  if (!cfg || !out) return fail(TTX_ERR_INVALID, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(TTX_ERR_NO_DEVICE, "no HIP device visible; libttx_hip has no CPU fallback");
  if (device < 0 || device >= n) return fail(TTX_ERR_INVALID, "device index out of range");
  if (!is_gfx950(device)) return fail(TTX_ERR_NO_DEVICE, "device is not gfx950 (MI355X); this library is built for gfx950 only");
  ttx_model* m = new ttx_model();
  m->cfg = *cfg;
  m->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) m->n_cu = prop.multiProcessorCount;
  }
  int r = build_layout(m);
  if (r != TTX_OK) { delete m; return r; }
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&m->blob, m->blob_floats * sizeof(float)) != hipSuccess) {
    delete m;
    return fail(TTX_ERR_NOMEM, "hipMalloc of the weight blob failed");
  }
  *out = m;
  return TTX_OK;
}

extern "C" int ttx_model_create_empty(const ttx_config* cfg, int device, ttx_model** out) {
  return model_alloc(cfg, device, out);
}

extern "C" int ttx_model_blob(ttx_model* m, void** d_ptr, int64_t* bytes) {
  if (!m || !d_ptr || !bytes) return fail(TTX_ERR_INVALID, "null argument");
  *d_ptr = m->blob;
  *bytes = (int64_t)(m->blob_floats * sizeof(float));
  return TTX_OK;
}

extern "C" int ttx_model_create(const ttx_config* cfg, const ttx_tensor* tensors, int n_tensors, int device,
                                ttx_model** out) {
  if (!tensors || n_tensors <= 0) return fail(TTX_ERR_INVALID, "no tensors given");
  ttx_model* m = nullptr;
  TTX_TRY(model_alloc(cfg, device, &m));
  std::vector<float> host(m->blob_floats, 0.f);
  std::map<std::string, const ttx_tensor*> given;
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data) { ttx_model_destroy(m); return fail(TTX_ERR_INVALID, "tensor with null name/data"); }
    std::string nm = tensors[i].name;
    if (nm.rfind("model.", 0) == 0) nm = nm.substr(6);  // Lightning checkpoint prefix
    given[nm] = &tensors[i];
  }
  for (auto& kv : m->index) {
    const std::string& nm = kv.first;
    const bool derived = nm.rfind("derived.", 0) == 0;
    auto it = given.find(nm);
    if (it == given.end()) {
      if (derived || nm == "positional_encoding.pe") continue;
      ttx_model_destroy(m);
      return fail(TTX_ERR_INVALID, "missing tensor: " + nm);
    }
    if ((size_t)it->second->numel != kv.second.second) {
      ttx_model_destroy(m);
      return fail(TTX_ERR_INVALID, "tensor " + nm + " has " + std::to_string(it->second->numel) + " elements, expected " +
                                       std::to_string(kv.second.second));
    }
    std::memcpy(host.data() + kv.second.first, it->second->data, kv.second.second * sizeof(float));
  }
  const ttx_config& c = m->cfg;
  const size_t d = c.embedding_dim;
  if (given.find("positional_encoding.pe") == given.end()) {
    // embeddings.py:38-45 in fp32: row 0 zeros, row p+1 = sin/cos(p * exp(2i * (-ln 1e4 / d)))
    float* pe = host.data() + m->pe;
    const float cst = (float)(-std::log(10000.0) / (double)d);
    for (int p = 0; p < c.max_positions; ++p)
      for (size_t i = 0; i < d; i += 2) {
        const float div = expf((float)i * cst);
        const float arg = (float)p * div;
        pe[(size_t)(p + 1) * d + i] = sinf(arg);
        pe[(size_t)(p + 1) * d + i + 1] = cosf(arg);
      }
  }
  for (int l = 0; l < c.num_decoder_layers; ++l) {
    // rows d..3d of multihead_attn.in_proj_weight are the K and V projections (torch MHA packing)
    std::memcpy(host.data() + m->cross_kv_w + (size_t)l * 2 * d * d, host.data() + m->dec[l].ca_in_w + d * d, 2 * d * d * sizeof(float));
    std::memcpy(host.data() + m->cross_kv_b + (size_t)l * 2 * d, host.data() + m->dec[l].ca_in_b + d, 2 * d * sizeof(float));
  }
  if (hipMemcpy(m->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    ttx_model_destroy(m);
    return fail(TTX_ERR_HIP, "hipMemcpy of the weight blob failed");
  }
  *out = m;
  return TTX_OK;
}

extern "C" void ttx_model_destroy(ttx_model* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

// Growing a workspace must not stall the other sessions' streams: hipFree waits for the whole device, so the old
// allocation is parked here and released at the start of a later top-level call (or when a session is destroyed),
// when nothing of the previous call is in flight any more.  Work already enqueued keeps using the old allocation;
// whatever is enqueued after the growth uses the new one (workspaces carry no state across a growth).
static std::mutex g_retired_mu;
static std::vector<void*> g_retired;

static void release_retired() {
  std::vector<void*> v;
  {
    std::lock_guard<std::mutex> lk(g_retired_mu);
    v.swap(g_retired);
  }
  if (v.empty()) return;
  (void)hipDeviceSynchronize();
  for (void* p : v) (void)hipFree(p);
}

// Session streams come from a per-device cache that outlives the sessions.  The runtime binds a stream to one of its few
// hardware queues when the stream is created, round-robin over every stream the process ever made: the streams of a SECOND
// model's sessions (after the first model was closed) landed two to a queue and its pools ran 7 % slower than in a fresh process
// (tools/exp/inline_order.py, profiles/r03_stream_reuse.txt).  Handing the same streams out again, lowest creation index first,
// gives every later set of sessions the queue layout of the first.
static std::mutex g_stream_mu;
static std::map<int, std::vector<std::pair<int, hipStream_t>>> g_free_streams;      // device -> (creation index, stream), idle
static std::map<hipStream_t, int> g_stream_index;
static int g_streams_made = 0;

static int ensure_own_stream(ttx_session* s) {
  if (s->own_stream) return TTX_OK;
  std::lock_guard<std::mutex> lock(g_stream_mu);
  auto& idle = g_free_streams[s->m->device];
  if (!idle.empty()) {
    auto it = std::min_element(idle.begin(), idle.end());
    s->own_stream = it->second;
    idle.erase(it);
    return TTX_OK;
  }
  HIP_TRY(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
  g_stream_index[s->own_stream] = g_streams_made++;
  return TTX_OK;
}

static void release_own_stream(ttx_session* s) {        // the stream is idle (the caller synchronised the device)
  if (!s->own_stream) return;
  std::lock_guard<std::mutex> lock(g_stream_mu);
  g_free_streams[s->m->device].push_back({g_stream_index[s->own_stream], s->own_stream});
  s->own_stream = nullptr;
}

static int ensure(Buf& b, size_t bytes, hipStream_t st) {
  (void)st;
  if (bytes <= b.cap) return TTX_OK;
  if (b.owner_gen) ++*b.owner_gen;
  if (b.p) {
    std::lock_guard<std::mutex> lk(g_retired_mu);
    g_retired.push_back(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  if (hipMalloc(&b.p, want) != hipSuccess) return fail(TTX_ERR_NOMEM, "hipMalloc failed for " + std::to_string(want) + " bytes");
  b.cap = want;
  // TTX_POISON_WORKSPACES=1 (tests): fresh workspaces are filled with 0xFF bytes (NaN as floats, -1 as ints), so that
  // any dependence on never-written workspace memory shows up deterministically instead of once in a while
  static const bool poison = getenv("TTX_POISON_WORKSPACES") != nullptr && atoi(getenv("TTX_POISON_WORKSPACES")) != 0;
  if (poison) {
    if (hipMemset(b.p, 0xFF, want) != hipSuccess) return fail(TTX_ERR_HIP, "hipMemset (poison) failed");
    (void)hipDeviceSynchronize();
  }
  return TTX_OK;
}

extern "C" int ttx_session_create(ttx_model* m, ttx_session** out) {
  if (!m || !out) return fail(TTX_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->device));
  ttx_session* s = new ttx_session();
  s->m = m;
  if (hipHostMalloc((void**)&s->host_info, sizeof(HostInfo), hipHostMallocMapped) != hipSuccess ||
      hipHostMalloc((void**)&s->host_state, sizeof(DecState), hipHostMallocDefault) != hipSuccess) {
    delete s;
    return fail(TTX_ERR_NOMEM, "hipHostMalloc failed");
  }
  std::memset(s->host_info, 0, sizeof(HostInfo));
  HIP_TRY(hipEventCreate(&s->ev_a));
  HIP_TRY(hipEventCreate(&s->ev_b));
  HIP_TRY(hipEventCreate(&s->ev_c));
  HIP_TRY(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
  s->use_graphs = getenv("TTX_NO_GRAPH") == nullptr;
  const char* pf = getenv("TTX_PROFILE_GEMM");
  s->profile = pf && pf[0] == '1';
  // tuning knob: the live row count below which a step takes the short-chain GEMM variant (identical bits)
  if (const char* e = getenv("TTX_SMALL_ROWS")) s->small_rows = std::max(0, atoi(e));
  if (const char* e = getenv("TTX_FFN2_SLAB_ROWS")) s->ffn2_slab_rows = std::max(0, atoi(e));
  if (const char* e = getenv("TTX_QKV_SMALL_ROWS")) s->qkv_small_rows = std::max(0, atoi(e));
  // test hook: every attention launch on the streaming fallback kernel
  if (const char* e = getenv("TTX_ATTN_FALLBACK")) s->attn_fallback = atoi(e) != 0;
  s->host_timing = getenv("TTX_HOST_TIMING") != nullptr;
  *out = s;
  return TTX_OK;
}

extern "C" void ttx_session_destroy(ttx_session* s) {
  if (!s) return;
  if (s->host_timing && s->host_launches)
    fprintf(stderr, "[ttx host timing] hipGraphLaunch: %lld launches, %.1f us each\n", s->host_launches,
            s->host_launch_us / (double)s->host_launches);
  if (s->host_timing && s->host_captures)
    fprintf(stderr, "[ttx host timing] beam iteration graphs captured: %lld, %.1f us each\n", s->host_captures,
            s->host_capture_us / (double)s->host_captures);
  if (s->dead) {        // hipFree / hipDeviceSynchronize would wait for the stuck stream, and a late kernel may still write
    delete s;           // the mapped host words: leak workspaces, pinned memory, graphs and events of a hung session
    return;
  }
  release_retired();
  (void)hipDeviceSynchronize();
  for (Buf* b : s->all)
    if (b->p) (void)hipFree(b->p);
  if (s->host_info) (void)hipHostFree(s->host_info);
  if (s->beam_host) (void)hipHostFree(s->beam_host);
  if (s->bp_host) (void)hipHostFree(s->bp_host);
  s->drop_graphs();
  release_own_stream(s);
  if (s->ev_done) (void)hipEventDestroy(s->ev_done);
  if (s->host_state) (void)hipHostFree(s->host_state);
  for (auto& e : s->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (s->ev_a) (void)hipEventDestroy(s->ev_a);
  if (s->ev_b) (void)hipEventDestroy(s->ev_b);
  if (s->ev_c) (void)hipEventDestroy(s->ev_c);
  delete s;
}

// ------------------------------------------------------------------------------------------------
// GEMM variant of a launch (ttx_internal.h: all variants return identical bits).  Steps: by live rows; bulk passes
// (encoder, cross K/V, full-prefix decoder) by their row count.
static int variant_for_rows(const ttx_session* s, long long rows, bool step) {
  if (!step) return GV_BIG;
  if (rows < s->qkv_small_rows) return GV_SMALL;
  if (rows < s->small_rows) return GV_MID;
  return rows < s->ffn2_slab_rows ? GV_BIG_FFN2_SLABS : GV_BIG;
}
// the step policy per GEMM shape (ttx_internal.h): QKV (N = 3d) | the d-wide K = d GEMMs and the classifier | FFN1 | FFN2
static int variant_qkv(int v) { return v == GV_SMALL ? GV_SMALL : GV_BIG; }
static int variant_dd(int v) { return (v == GV_SMALL || v == GV_MID) ? GV_SMALL : GV_BIG; }
static int variant_ffn1(int v) { return v == GV_SMALL ? GV_SMALL : GV_BIG; }
static int variant_ffn2(int v) { return v == GV_BIG ? GV_BIG : GV_SMALL; }

// d-wide GEMM -> slab(s) -> k_finish_ln: Y = LN2?(LN((resid + bias) + X W^T))
static int gemm_ln(ttx_session* s, hipStream_t st, const float* X, int ldx, int K, const float* W, const float* bias,
                   const float* resid, const float* g1, const float* b1, const float* g2, const float* b2,
                   const uint8_t* row_valid, float* Y, const int* m_ptr, int Mmax, int variant) {
  const int d = s->m->cfg.embedding_dim;
  const int S = gemm_splits(d, K, m_ptr != nullptr, variant);
  const long long stride = (long long)Mmax * d;
  TTX_TRY(ensure(s->slab, sizeof(float) * (size_t)S * stride, st));
  TTX_TRY(launch_gemm(s, st, X, ldx, W, K, nullptr, s->slab.as<float>(), d, m_ptr, Mmax, d, K, false, S, stride, variant));
  return launch_finish(s, st, s->slab.as<float>(), S, stride, bias, resid, g1, b1, g2, b2, row_valid, Y, m_ptr, Mmax);
}

static int ensure_acts(ttx_session* s, hipStream_t st, size_t M, int qkv_layers) {
  const ttx_config& c = s->m->cfg;
  const size_t d = c.embedding_dim, F = c.feedforward_dim;
  TTX_TRY(ensure(s->x, M * d * 4, st));
  TTX_TRY(ensure(s->x1, M * d * 4, st));
  TTX_TRY(ensure(s->x2, M * d * 4, st));
  TTX_TRY(ensure(s->xf, M * d * 4, st));
  TTX_TRY(ensure(s->ao, M * d * 4, st));
  TTX_TRY(ensure(s->q2, M * d * 4, st));
  TTX_TRY(ensure(s->hbuf, M * F * 4, st));
  TTX_TRY(ensure(s->qkv, (size_t)qkv_layers * M * 3 * d * 4, st));
  return TTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Encoder: tokens int32 [B*Ls] + valid mask -> memory [B*Ls, d] (zeros at PAD rows).  modules.py:110-116
// `qkv_buf`: where the encoder keeps its packed Q/K/V rows (default: the session's step buffer — callers whose NEXT step still
// needs the previous step's Q/K/V rows, i.e. the beam source pool's deferred cache hand-over, pass a buffer of their own).
static int run_encoder(ttx_session* s, hipStream_t st, const int* tok, const uint8_t* valid, int B, int Ls, float* memory,
                       float* qkv_buf = nullptr) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, F = c.feedforward_dim, H = c.num_heads;
  const int M = B * Ls;
  const int gv = variant_for_rows(s, M, false);
  TTX_TRY(ensure_acts(s, st, (size_t)M, 1));
  float* x = s->x.as<float>();
  float* x1 = s->x1.as<float>();
  float* qkv = qkv_buf ? qkv_buf : s->qkv.as<float>();
  float* ao = s->ao.as<float>();
  float* hb = s->hbuf.as<float>();
  EmbedArgs e{};
  e.table = m->p(m->src_emb); e.pe = m->p(m->pe); e.X = x; e.d = d; e.V = c.src_vocab_size; e.tok = tok; e.rows = M; e.L = Ls;
  hipLaunchKernelGGL((k_embed<false>), dim3(cdiv(M, 4)), dim3(256), 0, st, e);
  HIP_TRY(hipGetLastError());
  for (int l = 0; l < c.num_encoder_layers; ++l) {
    const LayerW& w = m->enc[l];
    const bool last = (l == c.num_encoder_layers - 1);
    TTX_TRY(launch_gemm(s, st, x, d, m->p(w.sa_in_w), d, m->p(w.sa_in_b), qkv, 3 * d, nullptr, M, 3 * d, d, false, 0, 0, gv));
    AttnArgs a{};
    a.q = qkv; a.ldq = 3 * d; a.k = qkv + d; a.v = qkv + 2 * d; a.ldkv = 3 * d; a.out = ao; a.d = d;
    a.scale = 1.0f / sqrtf((float)ATT_DH); a.L = Ls; a.tok = tok; a.pad = c.pad_token;
    TTX_TRY(launch_attn(ATT_ENC, s, st, a, B, H, Ls, Ls));
    TTX_TRY(gemm_ln(s, st, ao, d, d, m->p(w.sa_out_w), m->p(w.sa_out_b), x, m->p(w.n1_w), m->p(w.n1_b), nullptr, nullptr,
                    nullptr, x1, nullptr, M, gv));
    TTX_TRY(launch_gemm(s, st, x1, d, m->p(w.l1_w), d, m->p(w.l1_b), hb, F, nullptr, M, F, d, true, 0, 0, gv));
    TTX_TRY(gemm_ln(s, st, hb, F, F, m->p(w.l2_w), m->p(w.l2_b), x1, m->p(w.n2_w), m->p(w.n2_b),
                    last ? m->p(m->enc_norm_w) : nullptr, last ? m->p(m->enc_norm_b) : nullptr, last ? valid : nullptr,
                    last ? memory : x, nullptr, M, gv));
  }
  return TTX_OK;
}

static int prepare_tokens(hipStream_t st, const int64_t* d_in, int* out, uint8_t* valid, int n, int pad) {
  hipLaunchKernelGGL(k_prepare_tokens, dim3(cdiv(n, 256)), dim3(256), 0, st, d_in, out, valid, n, pad);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

extern "C" int ttx_encode_src(ttx_session* s, const int64_t* d_src, int B, int Ls, float* d_memory, void* stream) {
  if (!s || !d_src || !d_memory || B <= 0 || Ls <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_encode_src");
  if (Ls > s->m->cfg.max_positions) return fail(TTX_ERR_INVALID, "source longer than the positional table");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->m->device));
  TTX_TRY(ensure(s->tok_src, (size_t)B * Ls * 4, st));
  TTX_TRY(ensure(s->src_valid, (size_t)B * Ls, st));
  TTX_TRY(prepare_tokens(st, d_src, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B * Ls, s->m->cfg.pad_token));
  return run_encoder(s, st, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B, Ls, d_memory);
}

// ------------------------------------------------------------------------------------------------
// Full-prefix decoder (modules.py:118-138).  tok int32 [R*Lt]; memory fp32 [Rm*Ls, d]; mem_pad u8 [Rm*Ls].
static int run_decoder_full(ttx_session* s, hipStream_t st, const int* tok, int R, int Lt, const float* memory,
                            const uint8_t* mem_pad, const int* mem_row, int Rm, int Ls, float* logits) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, F = c.feedforward_dim, H = c.num_heads, V = c.vocab_size;
  const int M = R * Lt, Mk = Rm * Ls;
  const int gv = variant_for_rows(s, M, false);
  TTX_TRY(ensure_acts(s, st, (size_t)M, 1));
  TTX_TRY(ensure(s->ckv, (size_t)Mk * 2 * d * 4, st));
  float* x = s->x.as<float>();
  float* x1 = s->x1.as<float>();
  float* x2 = s->x2.as<float>();
  float* xf = s->xf.as<float>();
  float* qkv = s->qkv.as<float>();
  float* ao = s->ao.as<float>();
  float* q2 = s->q2.as<float>();
  float* hb = s->hbuf.as<float>();
  float* ckv = s->ckv.as<float>();
  const float scale = 1.0f / sqrtf((float)ATT_DH);
  EmbedArgs e{};
  e.table = m->p(m->tgt_emb); e.pe = m->p(m->pe); e.X = x; e.d = d; e.V = c.vocab_size; e.tok = tok; e.rows = M; e.L = Lt;
  hipLaunchKernelGGL((k_embed<false>), dim3(cdiv(M, 4)), dim3(256), 0, st, e);
  HIP_TRY(hipGetLastError());
  for (int l = 0; l < c.num_decoder_layers; ++l) {
    const LayerW& w = m->dec[l];
    const bool last = (l == c.num_decoder_layers - 1);
    TTX_TRY(launch_gemm(s, st, x, d, m->p(w.sa_in_w), d, m->p(w.sa_in_b), qkv, 3 * d, nullptr, M, 3 * d, d, false, 0, 0, gv));
    AttnArgs a{};
    a.q = qkv; a.ldq = 3 * d; a.k = qkv + d; a.v = qkv + 2 * d; a.ldkv = 3 * d; a.out = ao; a.d = d; a.scale = scale;
    a.L = Lt; a.tok = tok; a.pad = c.pad_token;
    TTX_TRY(launch_attn(ATT_FULL_SELF, s, st, a, R, H, Lt, Lt));
    TTX_TRY(gemm_ln(s, st, ao, d, d, m->p(w.sa_out_w), m->p(w.sa_out_b), x, m->p(w.n1_w), m->p(w.n1_b), nullptr, nullptr,
                    nullptr, x1, nullptr, M, gv));
    // cross attention: Q from the decoder stream, K/V re-projected from `memory` (as the reference does per call)
    TTX_TRY(launch_gemm(s, st, x1, d, m->p(w.ca_in_w), d, m->p(w.ca_in_b), q2, d, nullptr, M, d, d, false, 0, 0, gv));
    TTX_TRY(launch_gemm(s, st, memory, d, m->p(w.ca_in_w) + (size_t)d * d, d, m->p(w.ca_in_b) + d, ckv, 2 * d, nullptr, Mk,
                        2 * d, d, false, 0, 0, gv));
    AttnArgs ca{};
    ca.q = q2; ca.ldq = d; ca.k = ckv; ca.v = ckv + d; ca.ldkv = 2 * d; ca.out = ao; ca.d = d; ca.scale = scale;
    ca.L = Lt; ca.Lk = Ls; ca.key_pad = mem_pad; ca.mem_row = mem_row;
    TTX_TRY(launch_attn(ATT_FULL_CROSS, s, st, ca, R, H, Lt, Ls));
    TTX_TRY(gemm_ln(s, st, ao, d, d, m->p(w.ca_out_w), m->p(w.ca_out_b), x1, m->p(w.n2_w), m->p(w.n2_b), nullptr, nullptr,
                    nullptr, x2, nullptr, M, gv));
    TTX_TRY(launch_gemm(s, st, x2, d, m->p(w.l1_w), d, m->p(w.l1_b), hb, F, nullptr, M, F, d, true, 0, 0, gv));
    TTX_TRY(gemm_ln(s, st, hb, F, F, m->p(w.l2_w), m->p(w.l2_b), x2, m->p(w.n3_w), m->p(w.n3_b),
                    last ? m->p(m->dec_norm_w) : nullptr, last ? m->p(m->dec_norm_b) : nullptr, nullptr, last ? xf : x, nullptr, M, gv));
  }
  return launch_gemm(s, st, xf, d, m->p(m->cls_w), d, m->p(m->cls_b), logits, V, nullptr, M, V, d, false, 0, 0, gv);
}

extern "C" int ttx_decode_tgt(ttx_session* s, const int64_t* d_tgt, int R, int Lt, const float* d_memory,
                              const uint8_t* d_mem_pad, const int32_t* d_mem_row, int Rm, int Ls, float* d_logits,
                              void* stream) {
  if (!s || !d_tgt || !d_memory || !d_mem_pad || !d_logits || R <= 0 || Lt <= 0 || Rm <= 0 || Ls <= 0)
    return fail(TTX_ERR_INVALID, "bad argument to ttx_decode_tgt");
  if (!d_mem_row && Rm != R) return fail(TTX_ERR_INVALID, "without a row map the memory must have one row per decoder row");
  if (Lt > s->m->cfg.max_positions) return fail(TTX_ERR_INVALID, "target longer than the positional table");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->m->device));
  TTX_TRY(ensure(s->tok_tgt, (size_t)R * Lt * 4, st));
  TTX_TRY(prepare_tokens(st, d_tgt, s->tok_tgt.as<int>(), nullptr, R * Lt, s->m->cfg.pad_token));
  return run_decoder_full(s, st, s->tok_tgt.as<int>(), R, Lt, d_memory, d_mem_pad, d_mem_row, Rm, Ls, d_logits);
}

__global__ void k_invert_mask(const uint8_t* valid, uint8_t* pad, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pad[i] = valid[i] ? 0 : 1;
}

extern "C" int ttx_forward(ttx_session* s, const int64_t* d_src, int B, int Ls, const int64_t* d_tgt, int Lt,
                           float* d_logits, void* stream) {
  if (!s || !d_src || !d_tgt || !d_logits || B <= 0 || Ls <= 0 || Lt <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_forward");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->m->device));
  const int d = s->m->cfg.embedding_dim;
  TTX_TRY(ensure(s->memory, (size_t)B * Ls * d * 4, st));
  TTX_TRY(ensure(s->mem_pad_tmp, (size_t)B * Ls, st));
  TTX_TRY(ttx_encode_src(s, d_src, B, Ls, s->memory.as<float>(), stream));
  hipLaunchKernelGGL(k_invert_mask, dim3(cdiv(B * Ls, 256)), dim3(256), 0, st, s->src_valid.as<uint8_t>(),
                     s->mem_pad_tmp.as<uint8_t>(), B * Ls);
  HIP_TRY(hipGetLastError());
  return ttx_decode_tgt(s, d_tgt, B, Lt, s->memory.as<float>(), s->mem_pad_tmp.as<uint8_t>(), nullptr, B, Ls, d_logits, stream);
}

// ------------------------------------------------------------------------------------------------
static int clamp_draft_len(int draft_len, int lo, int hi) { return std::min(std::max(lo, draft_len), hi); }

template <typename OutT>
static int launch_make_drafts(hipStream_t st, const int* src32, int src_ld, int off, int B, int L, int N, int D, int eos,
                              int pad, int repl, OutT* out) {
  const int need = N + D - 1;
  const int Lp = L > need ? L : need;
  const size_t lds = sizeof(int) * (size_t)(Lp + 1);
  if (lds > 60 * 1024) return fail(TTX_ERR_INVALID, "make_drafts: padded source row too long");
  hipLaunchKernelGGL((k_make_drafts<OutT>), dim3(B), dim3(64), lds, st, src32, src_ld, off, L, N, D, eos, pad, repl, out);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

extern "C" int ttx_make_drafts(ttx_session* s, const int64_t* d_src, int B, int L, int draft_len, int n_drafts,
                               int min_draft_len, int max_draft_len, int eos_token, int pad_token, int replace_token,
                               int64_t* d_drafts, void* stream) {
  if (!s || !d_src || !d_drafts || B <= 0 || L <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_make_drafts");
  // the reference's assertions (drafting.py:39-43)
  if (n_drafts <= 0) return fail(TTX_ERR_REFERENCE, "The number of drafts must be greater than 0");
  if (min_draft_len > max_draft_len) return fail(TTX_ERR_REFERENCE, "The minimum draft length must not be greater than the maximum draft length");
  if (pad_token == replace_token || eos_token == replace_token || eos_token == pad_token)
    return fail(TTX_ERR_REFERENCE, "pad, eos and replace tokens must be pairwise different");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->m->device));
  const int D = clamp_draft_len(draft_len, min_draft_len, max_draft_len);
  if (D <= 0) return fail(TTX_ERR_INVALID, "draft length must be positive");
  TTX_TRY(ensure(s->src32, (size_t)B * L * 4, st));
  TTX_TRY(prepare_tokens(st, d_src, s->src32.as<int>(), nullptr, B * L, pad_token));
  return launch_make_drafts<int64_t>(st, s->src32.as<int>(), L, 0, B, L, n_drafts, D, eos_token, pad_token, replace_token, d_drafts);
}

// ------------------------------------------------------------------------------------------------
// One verify step of the greedy-speculative loop: D+1 new positions per (running row, draft) through the
// decoder with the KV cache, argmax, accept/retire, K/V commit.  Every kernel sizes its work from the
// device-resident DecState, so the launch sequence is identical for every step (graph-replayable).
struct StepCtx {
  int B, Ls, N, D, Lc, gen_ld, max_len;
  ttx_gen_params p;
  const float* kcache = nullptr;   // null: the session's greedy-path caches
  const float* vcache = nullptr;
  const int* src_of = nullptr;     // running row -> source row (tree decoding)
  const int* src_len = nullptr;    // slot pool: source keys per slot
  const int* cache_slot = nullptr; // batch pool: running row (candidate) -> slot of its KV cache
  bool want_argmax = true;
  int variant = GV_BIG;            // GemmVariant of this step's launches (bit-identical results; chosen from the live row count)
};

static int run_step(ttx_session* s, hipStream_t st, const StepCtx& k, int kcap) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, F = c.feedforward_dim, H = c.num_heads, V = c.vocab_size, Ld = c.num_decoder_layers;
  const int D1 = k.D + 1;
  const int RPS = step_rps(k.N, k.D);
  const int Mmax = k.B * RPS;
  const int vq = variant_qkv(k.variant), vd = variant_dd(k.variant), v1 = variant_ffn1(k.variant), vf = variant_ffn2(k.variant);
  DecState* dst = s->state.as<DecState>();
  const int* m_ptr = &dst->m_rows;
  float* x = s->x.as<float>();
  float* x1 = s->x1.as<float>();
  float* x2 = s->x2.as<float>();
  float* xf = s->xf.as<float>();
  float* ao = s->ao.as<float>();
  float* q2 = s->q2.as<float>();
  float* hb = s->hbuf.as<float>();
  float* logits = s->logits.as<float>();
  const float scale = 1.0f / sqrtf((float)ATT_DH);
  const long long qkv_layer = (long long)Mmax * 3 * d;
  const long long cache_seq = (long long)k.Lc * d;
  const long long cache_layer = (long long)k.B * cache_seq;

  EmbedArgs e{};
  e.table = m->p(m->tgt_emb); e.pe = m->p(m->pe); e.X = x; e.d = d; e.V = c.vocab_size;
  e.st = dst; e.act_idx = s->act_idx.as<int>(); e.front = s->front.as<int>(); e.gen = s->gen.as<int>(); e.gen_ld = k.gen_ld;
  e.drafts = s->drafts.as<int>(); e.N = k.N; e.D = k.D;
  hipLaunchKernelGGL((k_embed<true>), dim3(cdiv(Mmax, 4)), dim3(256), 0, st, e);
  HIP_TRY(hipGetLastError());

  for (int l = 0; l < Ld; ++l) {
    const LayerW& w = m->dec[l];
    const bool last = (l == Ld - 1);
    float* qkv = s->qkv.as<float>() + (size_t)l * qkv_layer;
    TTX_TRY(launch_gemm(s, st, x, d, m->p(w.sa_in_w), d, m->p(w.sa_in_b), qkv, 3 * d, m_ptr, Mmax, 3 * d, d, false, 0, 0, vq));
    AttnArgs a{};
    a.q = qkv; a.ldq = 3 * d; a.k = qkv + d; a.v = qkv + 2 * d; a.ldkv = 3 * d; a.out = ao; a.d = d; a.scale = scale;
    a.tok = s->gen.as<int>(); a.pad = c.pad_token; a.st = dst; a.act_idx = s->act_idx.as<int>(); a.front = s->front.as<int>();
    a.kcache = (k.kcache ? k.kcache : s->kcache.as<float>()) + (size_t)l * cache_layer;
    a.vcache = (k.vcache ? k.vcache : s->vcache.as<float>()) + (size_t)l * cache_layer;
    a.cache_seq_stride = cache_seq; a.gen_ld = k.gen_ld; a.N = k.N; a.D = k.D; a.cache_slot = k.cache_slot;
    TTX_TRY(launch_attn(ATT_STEP_SELF, s, st, a, k.B, H, RPS, kcap, k.N, D1));
    TTX_TRY(gemm_ln(s, st, ao, d, d, m->p(w.sa_out_w), m->p(w.sa_out_b), x, m->p(w.n1_w), m->p(w.n1_b), nullptr, nullptr,
                    nullptr, x1, m_ptr, Mmax, vd));
    TTX_TRY(launch_gemm(s, st, x1, d, m->p(w.ca_in_w), d, m->p(w.ca_in_b), q2, d, m_ptr, Mmax, d, d, false, 0, 0, vd));
    AttnArgs ca{};
    ca.q = q2; ca.ldq = d; ca.k = s->memkv.as<float>() + (size_t)l * 2 * d; ca.v = ca.k + d; ca.ldkv = Ld * 2 * d;
    ca.out = ao; ca.d = d; ca.scale = scale; ca.Lk = k.Ls; ca.key_pad = s->src_valid.as<uint8_t>();
    ca.st = dst; ca.act_idx = s->act_idx.as<int>(); ca.front = s->front.as<int>(); ca.N = k.N; ca.D = k.D;
    ca.src_of = k.src_of; ca.src_len = k.src_len;
    TTX_TRY(launch_attn(ATT_STEP_CROSS, s, st, ca, k.B, H, RPS, k.Ls, k.N, D1));
    TTX_TRY(gemm_ln(s, st, ao, d, d, m->p(w.ca_out_w), m->p(w.ca_out_b), x1, m->p(w.n2_w), m->p(w.n2_b), nullptr, nullptr,
                    nullptr, x2, m_ptr, Mmax, vd));
    TTX_TRY(launch_gemm(s, st, x2, d, m->p(w.l1_w), d, m->p(w.l1_b), hb, F, m_ptr, Mmax, F, d, true, 0, 0, v1));
    TTX_TRY(gemm_ln(s, st, hb, F, F, m->p(w.l2_w), m->p(w.l2_b), x2, m->p(w.n3_w), m->p(w.n3_b),
                    last ? m->p(m->dec_norm_w) : nullptr, last ? m->p(m->dec_norm_b) : nullptr, nullptr, last ? xf : x, m_ptr, Mmax, vf));
  }
  TTX_TRY(launch_gemm(s, st, xf, d, m->p(m->cls_w), d, m->p(m->cls_b), logits, V, m_ptr, Mmax, V, d, false, 0, 0, vd));
  if (k.want_argmax) {
    hipLaunchKernelGGL(k_argmax, dim3(cdiv(Mmax, 4)), dim3(256), 0, st, logits, V, s->pred.as<int>(), m_ptr, Mmax);
    HIP_TRY(hipGetLastError());
  }
  return TTX_OK;
}

struct GenCtx {
  StepCtx k;
  LoopArgs la;
  KvCopyArgs kc;
};

static int launch_accept_and_commit(ttx_session* s, hipStream_t st, const GenCtx& g, bool greedy) {
  if (greedy) hipLaunchKernelGGL(k_greedy_accept, dim3(1), dim3(256), 0, st, g.la);
  else hipLaunchKernelGGL(k_accept, dim3(1), dim3(g.k.B > 256 ? ACCEPT_THREADS : 256), 0, st, g.la);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_kvcopy, dim3(g.k.B, s->m->cfg.num_decoder_layers), dim3(256), 0, st, g.kc);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

// The polling loops give up after this long without a published step (a failed kernel never publishes).
static bool watchdog_expired(std::chrono::steady_clock::time_point since) {
  static const int limit_s = [] { const char* e = getenv("TTX_WATCHDOG_SECONDS"); return e && atoi(e) > 0 ? atoi(e) : 120; }();
  return std::chrono::steady_clock::now() - since > std::chrono::seconds(limit_s);
}

// A verify step did not publish within the watchdog time.  The stream may be stuck for good, so NOTHING here (or in the
// callers, on this path) synchronises it: the session is marked unusable, every later call on it fails at once, and
// destroying it leaks its workspaces instead of waiting for the device.  The process should report the error and exit
// non-zero (or continue in a fresh child process); it must not re-exec itself after having touched the GPU.
static int session_hung(ttx_session* s) {
  s->dead = true;
  return fail(TTX_ERR_HIP, "verify step did not publish its result within the watchdog time; the session is unusable "
                           "(its stream was NOT synchronised) - exit the process or continue in a fresh child process");
}

static int session_alive(const ttx_session* s) {
  if (s && s->dead) return fail(TTX_ERR_HIP, "this session hung in an earlier call and is unusable");
  return TTX_OK;
}

struct EventGuard {              // destroys the event on every exit path
  hipEvent_t e = nullptr;
  ~EventGuard() { if (e) (void)hipEventDestroy(e); }
};

// One generate call in flight on one session: start (encoder, drafts, loop init) -> steps -> finish.
struct GenJob {
  ttx_session* s = nullptr;
  hipStream_t st = nullptr;
  GenCtx g{};
  bool greedy = false;
  int64_t* d_out = nullptr;
  int16_t* d_traj = nullptr;     // per-row rule only: [B][max_len + 1] fronts after every step (-1 past the row's last step)
  int32_t* d_fin = nullptr;      // per-row rule only: [B] step at which the row produced EOS (0: never)
  ttx_gen_stats* stats = nullptr;
  int launched = 0;
  int phase = 0;          // 0 idle, 1 running, 2 finishing
  int batch = -1;
  unsigned idle_spins = 0;
  std::chrono::steady_clock::time_point last_progress = std::chrono::steady_clock::now();
};

static int gen_validate(const ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_gen_params* p, const int64_t* d_out,
                        bool greedy) {
  if (!s || !d_src || !p || !d_out || B <= 0 || Ls <= 1) return fail(TTX_ERR_INVALID, "bad argument to a generate call");
  const ttx_config& c = s->m->cfg;
  if (!greedy) {
    if (p->n_drafts <= 0) return fail(TTX_ERR_REFERENCE, "The number of drafts must be greater than 0");
    if (p->max_len < 1) return fail(TTX_ERR_REFERENCE, "The minimum draft length must not be greater than the maximum draft length");
    if (p->pad_token == p->replace_token || p->eos_token == p->replace_token || p->eos_token == p->pad_token)
      return fail(TTX_ERR_REFERENCE, "pad, eos and replace tokens must be pairwise different");
    if (p->draft_len <= 0) return fail(TTX_ERR_REFERENCE, "Number of speculative tokens must be a positive integer.");
    if (p->draft_len > p->max_len) return fail(TTX_ERR_REFERENCE, "draft_len beyond max_len: the reference's scatter shapes disagree");
  } else if (p->max_len < 1) {
    return fail(TTX_ERR_INVALID, "max_len must be positive");
  }
  if (p->pad_token != c.pad_token) return fail(TTX_ERR_INVALID, "generator pad token differs from the model's");
  const int D = greedy ? 0 : p->draft_len;
  if (p->max_len + D + 2 > c.max_positions) return fail(TTX_ERR_INVALID, "max_len + draft_len exceeds the positional table");
  return TTX_OK;
}

static int gen_start(GenJob& j, ttx_session* s, hipStream_t st, const int64_t* d_src, int B, int Ls, const ttx_gen_params* p,
                     int64_t* d_out, ttx_gen_stats* stats, bool greedy, int16_t* d_traj = nullptr, int32_t* d_fin = nullptr) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers, V = c.vocab_size;
  const int N = greedy ? 1 : p->n_drafts, D = greedy ? 0 : p->draft_len, D1 = D + 1;
  const int max_len = p->max_len;
  j.s = s; j.st = st; j.greedy = greedy; j.d_out = d_out; j.stats = stats; j.launched = 0;
  j.d_traj = d_traj; j.d_fin = d_fin;
  const bool row_rule = d_traj != nullptr;
  s->snap_step = 0;
  GenCtx& g = j.g;
  g = GenCtx{};
  g.k.B = B; g.k.Ls = Ls; g.k.N = N; g.k.D = D; g.k.max_len = max_len; g.k.p = *p;
  g.k.Lc = max_len + D1;           // cache positions per row (front + D < max_len + D)
  g.k.gen_ld = max_len + D + 2;
  const size_t Mmax = (size_t)B * step_rps(N, D);

  TTX_TRY(ensure(s->tok_src, (size_t)B * Ls * 4, st));
  TTX_TRY(ensure(s->src_valid, (size_t)B * Ls, st));
  TTX_TRY(ensure(s->memory, (size_t)B * Ls * d * 4, st));
  TTX_TRY(ensure(s->memkv, (size_t)B * Ls * Ld * 2 * d * 4, st));
  TTX_TRY(ensure(s->drafts, (size_t)B * N * std::max(D, 1) * 4, st));
  TTX_TRY(ensure(s->gen, (size_t)B * g.k.gen_ld * 4, st));
  TTX_TRY(ensure(s->front, (size_t)B * 4, st));
  TTX_TRY(ensure(s->act_idx, (size_t)B * 4, st));
  TTX_TRY(ensure(s->haspad, (size_t)B * 4, st));
  if (row_rule) {
    TTX_TRY(ensure(s->traj, (size_t)B * (max_len + 1) * 2, st));
    TTX_TRY(ensure(s->fin_step, (size_t)B * 4, st));
  }
  TTX_TRY(ensure(s->rec, (size_t)B * sizeof(CopyRec), st));
  TTX_TRY(ensure(s->pred, Mmax * 4, st));
  TTX_TRY(ensure(s->state, sizeof(DecState), st));
  TTX_TRY(ensure(s->logits, Mmax * V * 4, st));
  TTX_TRY(ensure(s->outbuf, (size_t)B * max_len * 8, st));
  TTX_TRY(ensure(s->kcache, (size_t)Ld * B * g.k.Lc * d * 4, st));
  TTX_TRY(ensure(s->vcache, (size_t)Ld * B * g.k.Lc * d * 4, st));
  // the encoder and the step share the activation buffers; size them for the larger of the two
  const size_t Macts = std::max(Mmax, (size_t)B * Ls);
  TTX_TRY(ensure_acts(s, st, Macts, 1));
  TTX_TRY(ensure(s->qkv, std::max((size_t)Ld * Mmax, (size_t)B * Ls) * 3 * d * 4, st));
  TTX_TRY(ensure(s->slab, sizeof(float) * 16 * Macts * d, st));   // no allocation may happen inside a graph capture
  s->graphs_current();                                            // captured pointers may be stale

  s->ev_used = 0;
  HIP_TRY(hipEventRecord(s->ev_a, st));
  // encoder once per batch (:60-61) + cross-attention K/V of every decoder layer once per source
  TTX_TRY(prepare_tokens(st, d_src, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B * Ls, c.pad_token));
  TTX_TRY(run_encoder(s, st, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B, Ls, s->memory.as<float>()));
  const int gv = variant_for_rows(s, (long long)B * Ls, false);
  TTX_TRY(launch_gemm(s, st, s->memory.as<float>(), d, m->p(m->cross_kv_w), d, m->p(m->cross_kv_b), s->memkv.as<float>(),
                      Ld * 2 * d, nullptr, B * Ls, Ld * 2 * d, d, false, 0, 0, gv));
  // drafts from src[:, 1:] (:64-73): min_draft_len 1, max_draft_len max_len
  if (!greedy)
    TTX_TRY(launch_make_drafts<int>(st, s->tok_src.as<int>(), Ls, 1, B, Ls - 1, N, D, p->eos_token, p->pad_token,
                                    p->replace_token, s->drafts.as<int>()));

  g.la.st = s->state.as<DecState>(); g.la.act_idx = s->act_idx.as<int>(); g.la.front = s->front.as<int>();
  g.la.gen = s->gen.as<int>(); g.la.gen_ld = g.k.gen_ld; g.la.drafts = s->drafts.as<int>(); g.la.pred = s->pred.as<int>();
  g.la.rec = s->rec.as<CopyRec>(); g.la.out = s->outbuf.as<int64_t>(); g.la.haspad = s->haspad.as<int>();
  HostInfo* dev_info = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dev_info, (void*)s->host_info, 0));
  g.la.host = dev_info;
  g.la.B = B; g.la.N = N; g.la.D = D; g.la.Ls = Ls; g.la.max_len = max_len; g.la.pad = p->pad_token; g.la.bos = p->bos_token;
  g.la.eos = p->eos_token;
  g.la.row_rule = row_rule ? 1 : 0; g.la.traj = row_rule ? s->traj.as<short>() : nullptr; g.la.traj_ld = max_len + 1;
  g.la.fin_step = row_rule ? s->fin_step.as<int>() : nullptr;
  g.kc.st = s->state.as<DecState>(); g.kc.rec = s->rec.as<CopyRec>(); g.kc.qkv = s->qkv.as<float>();
  g.kc.qkv_layer_stride = (long long)Mmax * 3 * d;
  g.kc.kcache = s->kcache.as<float>(); g.kc.vcache = s->vcache.as<float>();
  g.kc.cache_seq_stride = (long long)g.k.Lc * d; g.kc.cache_layer_stride = (long long)B * g.k.Lc * d;
  g.kc.N = N; g.kc.D = D; g.kc.d = d;

  s->host_info->stop = 0;
  s->host_info->steps_done = 0;
  s->host_info->width = 1;
  s->host_info->n_active = B;
  hipLaunchKernelGGL(k_loop_init, dim3(64), dim3(256), 0, st, g.la);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev_b, st));
  j.phase = 1;
  return TTX_OK;
}

// Enqueue one verify step: replay the captured graph for this (shape, key-capacity bucket), capturing it on
// first use.  `width_bound` bounds the reference's generated width when the step runs.
static int gen_launch_step(GenJob& j, int width_bound) {
  ttx_session* s = j.s;
  StepCtx k = j.g.k;
  // GEMM variant from the rows this step will have (published by the previous step's accept kernel; plain greedy decoding
  // keeps all B rows): a free choice, every variant returns the same bits
  const int live = j.greedy ? k.B : std::max(1, (int)((volatile HostInfo*)s->host_info)->n_active);
  k.variant = variant_for_rows(s, (long long)live * step_rps(k.N, k.D), true);
  // prefix keys this step can see: < width_bound; bucket the LDS images of the self-attention in steps of 64 keys
  int kcap = std::min(k.max_len, ((std::max(width_bound, 1) + 63) / 64) * 64);
  const bool snapshot = (k.p.want_logits > 0 && k.p.want_logits == j.launched + 1);
  const bool use_graph = s->use_graphs && !s->profile && !snapshot;
  if (!use_graph) {
    TTX_TRY(run_step(s, j.st, k, kcap));
    if (snapshot) {
      // keep this step's pre-argmax logits and the loop state they belong to (before accept changes it)
      const ttx_config& c = s->m->cfg;
      const size_t Mmax = (size_t)k.B * step_rps(k.N, k.D);
      TTX_TRY(ensure(s->snap_logits, Mmax * c.vocab_size * 4, j.st));
      TTX_TRY(ensure(s->snap_act, (size_t)k.B * 4, j.st));
      TTX_TRY(ensure(s->snap_front, (size_t)k.B * 4, j.st));
      TTX_TRY(ensure(s->snap_gen, (size_t)k.B * k.gen_ld * 4, j.st));
      TTX_TRY(ensure(s->snap_state, sizeof(DecState), j.st));
      HIP_TRY(hipMemcpyAsync(s->snap_logits.p, s->logits.p, Mmax * c.vocab_size * 4, hipMemcpyDeviceToDevice, j.st));
      HIP_TRY(hipMemcpyAsync(s->snap_act.p, s->act_idx.p, (size_t)k.B * 4, hipMemcpyDeviceToDevice, j.st));
      HIP_TRY(hipMemcpyAsync(s->snap_front.p, s->front.p, (size_t)k.B * 4, hipMemcpyDeviceToDevice, j.st));
      HIP_TRY(hipMemcpyAsync(s->snap_gen.p, s->gen.p, (size_t)k.B * k.gen_ld * 4, hipMemcpyDeviceToDevice, j.st));
      HIP_TRY(hipMemcpyAsync(s->snap_state.p, s->state.p, sizeof(DecState), hipMemcpyDeviceToDevice, j.st));
      s->snap_B = k.B; s->snap_rps = step_rps(k.N, k.D); s->snap_gen_ld = k.gen_ld; s->snap_step = j.launched + 1;
    }
    TTX_TRY(launch_accept_and_commit(s, j.st, j.g, j.greedy));
    ++j.launched;
    return TTX_OK;
  }
  GraphKey key{k.B, k.Ls, k.N, k.D, k.max_len, j.greedy ? 1 : (j.g.la.row_rule ? 2 : 0), kcap, k.variant};
  s->graphs_current();
  auto it = s->graphs.find(key);
  if (it == s->graphs.end()) {
    if (!s->warmed.count(key)) {
      // first use of a shape runs eagerly once: function attributes (dynamic LDS limits) are set outside capture
      s->warmed.insert(key);
      TTX_TRY(run_step(s, j.st, k, kcap));
      TTX_TRY(launch_accept_and_commit(s, j.st, j.g, j.greedy));
      ++j.launched;
      return TTX_OK;
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(j.st, hipStreamCaptureModeThreadLocal));
    int rc = run_step(s, j.st, k, kcap);
    if (rc == TTX_OK) rc = launch_accept_and_commit(s, j.st, j.g, j.greedy);
    hipError_t e = hipStreamEndCapture(j.st, &graph);
    if (rc != TTX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    if (s->graphs.size() > 512) s->drop_graphs();
    it = s->graphs.emplace(key, exec).first;
  }
  if (s->host_timing) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipGraphLaunch(it->second, j.st));
    s->host_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    s->host_launches += 1;
  } else {
    HIP_TRY(hipGraphLaunch(it->second, j.st));
  }
  ++j.launched;
  return TTX_OK;
}

static int gen_finish_enqueue(GenJob& j) {
  ttx_session* s = j.s;
  const StepCtx& k = j.g.k;
  if (j.greedy) {
    hipLaunchKernelGGL(k_gen_to_out, dim3(cdiv(k.B * k.max_len, 256)), dim3(256), 0, j.st, s->gen.as<int>(), k.gen_ld, j.d_out,
                       k.B, k.max_len);
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemcpyAsync(j.d_out, s->outbuf.as<int64_t>(), (size_t)k.B * k.max_len * 8, hipMemcpyDeviceToDevice, j.st));
    if (j.d_traj) {
      HIP_TRY(hipMemcpyAsync(j.d_traj, s->traj.p, (size_t)k.B * (k.max_len + 1) * 2, hipMemcpyDeviceToDevice, j.st));
      HIP_TRY(hipMemcpyAsync(j.d_fin, s->fin_step.p, (size_t)k.B * 4, hipMemcpyDeviceToDevice, j.st));
    }
  }
  HIP_TRY(hipEventRecord(s->ev_c, j.st));
  HIP_TRY(hipMemcpyAsync(s->host_state, s->state.as<DecState>(), sizeof(DecState), hipMemcpyDeviceToHost, j.st));
  HIP_TRY(hipEventRecord(s->ev_done, j.st));
  j.phase = 2;
  return TTX_OK;
}

// GEMM event pairs of the work just finished on `st` -> the session's running sums (bench.py roofline).
static void collect_gemm_profile(ttx_session* s, hipStream_t st) {
  if (!s->profile) return;
  // summed over the generate calls since the last ttx_last_kernel_profile read
  s->prof_launches += (long long)s->ev_used;
  for (size_t i = 0; i < s->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev_pool[i].first, s->ev_pool[i].second) == hipSuccess) s->prof_ms += ms;
  }
  s->ev_used = 0;
  // What the bracketing itself adds to a launch's figure: pairs around a kernel of known duration (k_spin reports the
  // realtime ticks it saw go by), pair time minus in-kernel time, median of 32.
  if (s->prof_empty_pair_ms < 0 && s->ev_pool.size() >= 32) {
    unsigned long long* d_ticks = nullptr;
    if (hipMalloc(&d_ticks, 32 * sizeof(unsigned long long)) == hipSuccess) {
      for (int i = 0; i < 32; ++i) {
        (void)hipEventRecord(s->ev_pool[i].first, st);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, d_ticks + i, 2000);          // 20 us
        (void)hipEventRecord(s->ev_pool[i].second, st);
      }
      (void)hipStreamSynchronize(st);
      unsigned long long h_ticks[32];
      std::vector<double> over;
      if (hipMemcpy(h_ticks, d_ticks, sizeof(h_ticks), hipMemcpyDeviceToHost) == hipSuccess)
        for (int i = 0; i < 32; ++i) {
          float ms = 0.f;
          if (hipEventElapsedTime(&ms, s->ev_pool[i].first, s->ev_pool[i].second) == hipSuccess)
            over.push_back((double)ms - (double)h_ticks[i] * 1e-5);                        // 100 MHz ticks -> ms
        }
      (void)hipFree(d_ticks);
      if (!over.empty()) {
        std::sort(over.begin(), over.end());
        s->prof_empty_pair_ms = std::max(0.0, over[over.size() / 2]);
      }
    }
  }
}

static int gen_finish_collect(GenJob& j) {
  ttx_session* s = j.s;
  const DecState& hs = *s->host_state;
  if (j.stats) {
    j.stats->model_calls = hs.steps;
    j.stats->accepted_tokens = hs.accepted;
    j.stats->produced_tokens = hs.produced;
    j.stats->verified_positions = hs.verified_positions;
    j.stats->kv_prefix_positions = hs.kv_prefix_positions;
    j.stats->src_positions = hs.src_positions;
    j.stats->src_tokens_padded = (long long)j.g.k.B * j.g.k.Ls;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_a, s->ev_b));
    j.stats->encode_ms = ms;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_b, s->ev_c));
    j.stats->decode_ms = ms;
  }
  collect_gemm_profile(s, j.st);
  j.phase = 0;
  if (j.stats) j.stats->status = hs.error == 3 ? TTX_ERR_ROW_REPLAY : (hs.error ? TTX_ERR_REFERENCE : TTX_OK);
  if (hs.error == 3)
    return fail(TTX_ERR_ROW_REPLAY, "a row emitted PAD inside its sequence: what the reference does next depends on the other rows "
                                    "of its batch, so the batch must be decoded as given (ttx_greedy_speculative_generate_many)");
  if (hs.error == 2)
    return fail(TTX_ERR_REFERENCE, "the model emitted PAD inside a sequence and a whole column became PAD: the reference's draft "
                                   "scatter raises 'index out of range' here (speculative_decoding.py:97,111-115)");
  if (hs.error) return fail(TTX_ERR_REFERENCE, "a row finished at a width beyond max_len: shape mismatch in the reference (speculative_decoding.py:158)");
  return TTX_OK;
}

static int generate_common(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_gen_params* p, int64_t* d_out,
                           ttx_gen_stats* stats, void* stream, bool greedy) {
  TTX_TRY(gen_validate(s, d_src, B, Ls, p, d_out, greedy));
  TTX_TRY(session_alive(s));
  HIP_TRY(hipSetDevice(s->m->device));
  release_retired();
  // The loop runs on the session's own stream (the caller's may be the legacy null stream, which cannot be
  // captured into a graph); it first waits for the caller's stream, and the call returns only after the
  // session stream has drained, so the outputs are visible to whatever the caller enqueues next.
  TTX_TRY(ensure_own_stream(s));
  HIP_TRY(hipEventRecord(s->ev_done, (hipStream_t)stream));
  HIP_TRY(hipStreamWaitEvent(s->own_stream, s->ev_done, 0));
  hipStream_t st = s->own_stream;
  GenJob j;
  TTX_TRY(gen_start(j, s, st, d_src, B, Ls, p, d_out, stats, greedy));
  const int D1 = j.g.k.D + 1;
  // One step in flight: the next step is enqueued as soon as the accept kernel has published the previous one's
  // result to the host-mapped words (no stream synchronisation inside the loop).
  volatile HostInfo* hi = s->host_info;
  HIP_TRY(hipStreamSynchronize(st));            // loop init published (also covers max_len <= 1: stop already set)
  while (!hi->stop) {
    if (j.launched > p->max_len + 2) return fail(TTX_ERR_HIP, "decode loop failed to terminate");
    TTX_TRY(gen_launch_step(j, hi->width + D1));
    const int want = j.launched;
    unsigned spins = 0;
    const auto since = std::chrono::steady_clock::now();
    while (hi->steps_done < want && !hi->stop) {            // no HIP call while polling: only the clock
      if ((++spins & 0xffff) == 0 && watchdog_expired(since)) return session_hung(s);
      __builtin_ia32_pause();
    }
  }
  TTX_TRY(gen_finish_enqueue(j));
  HIP_TRY(hipStreamSynchronize(st));
  return gen_finish_collect(j);
}

extern "C" int ttx_greedy_speculative_generate(ttx_session* s, const int64_t* d_src, int B, int Ls,
                                               const ttx_gen_params* p, int64_t* d_out, ttx_gen_stats* stats,
                                               void* stream) {
  return generate_common(s, d_src, B, Ls, p, d_out, stats, stream, false);
}

extern "C" int ttx_greedy_generate(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_gen_params* p,
                                   int64_t* d_out, ttx_gen_stats* stats, void* stream) {
  return generate_common(s, d_src, B, Ls, p, d_out, stats, stream, true);
}

// Several batches in flight on one GPU: batch i is decoded on session i % n_sessions, each
// session on its own stream; one host thread round-robins over the sessions, enqueueing the next verify step of
// whichever session has published its previous one.  Outputs per batch are identical to the one-at-a-time call.
// ------------------------------------------------------------------------------------------------
// Slot pool: continuous batching under the per-row rule.  A session owns `C` slots; whenever enough of them are free
// the next rows of the (length-sorted) work list are encoded and admitted, so the verify step keeps close to
// C * (1 + N*D) rows from the first step to the last instead of decaying with every retiring sequence, and there is
// no tail of half-empty groups.  The step graph has one shape per session (C, Ls_cap): captured once.
struct PoolJob {
  ttx_session* s = nullptr;
  hipStream_t st = nullptr;
  GenCtx g{};
  int C = 0, Ls_cap = 0;
  int launched = 0;
  int phase = 0;            // 0 not started, 1 running, 2 finishing, 3 done
  int admits = 0;
  int quota_end = -1;       // serial (profiling) pass: end of this pool's share of the work list
  PoolIo io{};
  unsigned idle_spins = 0;
  std::chrono::steady_clock::time_point last_progress = std::chrono::steady_clock::now();
  long long admitted_rows = 0, src_tokens_padded = 0;
};

static int pool_start(PoolJob& j, ttx_session* s, hipStream_t st, int C, int Ls_cap, const ttx_gen_params* p, int64_t* d_out,
                      int16_t* d_traj, int32_t* d_fin) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers, V = c.vocab_size;
  const int N = p->n_drafts, D = p->draft_len, D1 = D + 1, max_len = p->max_len;
  j.s = s; j.st = st; j.C = C; j.Ls_cap = Ls_cap; j.launched = 0; j.admits = 0; j.admitted_rows = 0; j.src_tokens_padded = 0;
  s->snap_step = 0;
  GenCtx& g = j.g;
  g = GenCtx{};
  g.k.B = C; g.k.Ls = Ls_cap; g.k.N = N; g.k.D = D; g.k.max_len = max_len; g.k.p = *p;
  g.k.Lc = max_len + D1;
  g.k.gen_ld = max_len + D + 2;
  const size_t Mmax = (size_t)C * step_rps(N, D);
  const size_t kv_row = (size_t)Ld * 2 * d;
  TTX_TRY(ensure(s->tok_src, (size_t)C * Ls_cap * 4, st));
  TTX_TRY(ensure(s->valid_new, (size_t)C * Ls_cap, st));
  TTX_TRY(ensure(s->src_valid, (size_t)C * Ls_cap, st));
  TTX_TRY(ensure(s->memory, (size_t)C * Ls_cap * d * 4, st));
  TTX_TRY(ensure(s->memkv_new, (size_t)C * Ls_cap * kv_row * 4, st));
  TTX_TRY(ensure(s->memkv, (size_t)C * Ls_cap * kv_row * 4, st));
  TTX_TRY(ensure(s->drafts, (size_t)C * N * D * 4, st));
  TTX_TRY(ensure(s->drafts_new, (size_t)C * N * D * 4, st));
  TTX_TRY(ensure(s->gen, (size_t)C * g.k.gen_ld * 4, st));
  for (Buf* b : {&s->front, &s->act_idx, &s->haspad, &s->rstep, &s->row_of, &s->src_len, &s->new_slot}) TTX_TRY(ensure(*b, (size_t)C * 4, st));
  TTX_TRY(ensure(s->rec, (size_t)C * sizeof(CopyRec), st));
  TTX_TRY(ensure(s->pred, Mmax * 4, st));
  TTX_TRY(ensure(s->state, sizeof(DecState), st));
  TTX_TRY(ensure(s->pool_io, sizeof(PoolIo), st));
  TTX_TRY(ensure(s->logits, Mmax * V * 4, st));
  TTX_TRY(ensure(s->kcache, (size_t)Ld * C * g.k.Lc * d * 4, st));
  TTX_TRY(ensure(s->vcache, (size_t)Ld * C * g.k.Lc * d * 4, st));
  const size_t Macts = std::max(Mmax, (size_t)C * Ls_cap);
  TTX_TRY(ensure_acts(s, st, Macts, 1));
  TTX_TRY(ensure(s->qkv, std::max((size_t)Ld * Mmax, (size_t)C * Ls_cap) * 3 * d * 4, st));
  TTX_TRY(ensure(s->slab, sizeof(float) * 16 * Macts * d, st));
  s->graphs_current();

  s->ev_used = 0;
  HIP_TRY(hipEventRecord(s->ev_a, st));
  j.io = PoolIo{d_out, d_traj, d_fin, max_len + 1, 0};        // lives in the job until every stream has drained
  HIP_TRY(hipMemcpyAsync(s->pool_io.p, &j.io, sizeof(j.io), hipMemcpyHostToDevice, st));
  g.la.st = s->state.as<DecState>(); g.la.act_idx = s->act_idx.as<int>(); g.la.front = s->front.as<int>();
  g.la.gen = s->gen.as<int>(); g.la.gen_ld = g.k.gen_ld; g.la.drafts = s->drafts.as<int>(); g.la.pred = s->pred.as<int>();
  g.la.rec = s->rec.as<CopyRec>(); g.la.out = nullptr; g.la.haspad = s->haspad.as<int>();
  HostInfo* dev_info = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dev_info, (void*)s->host_info, 0));
  g.la.host = dev_info;
  g.la.B = C; g.la.N = N; g.la.D = D; g.la.Ls = Ls_cap; g.la.max_len = max_len; g.la.pad = p->pad_token; g.la.bos = p->bos_token;
  g.la.eos = p->eos_token;
  g.la.row_rule = 1; g.la.traj = nullptr; g.la.traj_ld = max_len + 1; g.la.fin_step = nullptr;
  g.la.pool = 1; g.la.rstep = s->rstep.as<int>(); g.la.row_of = s->row_of.as<int>(); g.la.io = s->pool_io.as<PoolIo>();
  g.k.src_len = s->src_len.as<int>();
  g.kc.st = s->state.as<DecState>(); g.kc.rec = s->rec.as<CopyRec>(); g.kc.qkv = s->qkv.as<float>();
  g.kc.qkv_layer_stride = (long long)Mmax * 3 * d;
  g.kc.kcache = s->kcache.as<float>(); g.kc.vcache = s->vcache.as<float>();
  g.kc.cache_seq_stride = (long long)g.k.Lc * d; g.kc.cache_layer_stride = (long long)C * g.k.Lc * d;
  g.kc.N = N; g.kc.D = D; g.kc.d = d;
  s->host_info->stop = 0; s->host_info->steps_done = 0; s->host_info->width = 1; s->host_info->n_active = 0;
  hipLaunchKernelGGL(k_pool_init, dim3(4), dim3(256), 0, st, g.la);
  HIP_TRY(hipGetLastError());
  j.phase = 1;
  return TTX_OK;
}

// Encode R new rows (rows first_row .. first_row+R-1 of the caller's sorted matrix, Ls_new columns of it) and hand
// them free slots.
static int pool_admit(PoolJob& j, const int64_t* d_src_rows, int ld_src, int R, int Ls_new, int first_row, int64_t* d_out,
                      int16_t* d_traj, int32_t* d_fin) {
  ttx_session* s = j.s;
  hipStream_t st = j.st;
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const StepCtx& k = j.g.k;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers;
  const int kv_row = Ld * 2 * d;
  hipLaunchKernelGGL(k_prepare_tokens_2d, dim3(cdiv(R * Ls_new, 256)), dim3(256), 0, st, d_src_rows, ld_src, s->tok_src.as<int>(),
                     s->valid_new.as<uint8_t>(), R, Ls_new, c.pad_token);
  HIP_TRY(hipGetLastError());
  TTX_TRY(run_encoder(s, st, s->tok_src.as<int>(), s->valid_new.as<uint8_t>(), R, Ls_new, s->memory.as<float>()));
  const int gv = variant_for_rows(s, (long long)R * Ls_new, false);
  TTX_TRY(launch_gemm(s, st, s->memory.as<float>(), d, m->p(m->cross_kv_w), d, m->p(m->cross_kv_b), s->memkv_new.as<float>(),
                      kv_row, nullptr, R * Ls_new, kv_row, d, false, 0, 0, gv));
  TTX_TRY(launch_make_drafts<int>(st, s->tok_src.as<int>(), Ls_new, 1, R, Ls_new - 1, k.N, k.D, k.p.eos_token, k.p.pad_token,
                                  k.p.replace_token, s->drafts_new.as<int>()));
  PoolAdmitArgs a{};
  a.st = s->state.as<DecState>(); a.act_idx = s->act_idx.as<int>(); a.front = s->front.as<int>(); a.haspad = s->haspad.as<int>();
  a.rstep = s->rstep.as<int>(); a.row_of = s->row_of.as<int>(); a.src_len = s->src_len.as<int>(); a.new_slot = s->new_slot.as<int>();
  a.host = j.g.la.host; a.B = j.C; a.N = k.N; a.D = k.D; a.R = R; a.first_row = first_row; a.Ls_new = Ls_new;
  hipLaunchKernelGGL(k_pool_admit, dim3(1), dim3(256), (size_t)j.C * 4, st, a);
  HIP_TRY(hipGetLastError());
  PoolFillArgs f{};
  f.new_slot = s->new_slot.as<int>(); f.R = R; f.gen = s->gen.as<int>(); f.gen_ld = k.gen_ld; f.bos = k.p.bos_token; f.pad = k.p.pad_token;
  f.drafts = s->drafts.as<int>(); f.drafts_new = s->drafts_new.as<int>(); f.nd = k.N * k.D;
  f.src_valid = s->src_valid.as<uint8_t>(); f.valid_new = s->valid_new.as<uint8_t>(); f.Ls_cap = j.Ls_cap; f.Ls_new = Ls_new;
  f.memkv = s->memkv.as<float>(); f.memkv_new = s->memkv_new.as<float>(); f.kv_row = kv_row;
  f.out_rows = d_out; f.traj_rows = d_traj; f.fin_rows = d_fin; f.max_len = k.max_len; f.traj_ld = k.max_len + 1; f.first_row = first_row;
  hipLaunchKernelGGL(k_pool_fill, dim3(R, 1 + Ls_new), dim3(256), 0, st, f);
  HIP_TRY(hipGetLastError());
  ++j.admits;
  j.admitted_rows += R;
  j.src_tokens_padded += (long long)R * Ls_new;
  return TTX_OK;
}

static int pool_launch_step(PoolJob& j, int n_live) {
  ttx_session* s = j.s;
  StepCtx k = j.g.k;
  k.variant = variant_for_rows(s, (long long)n_live * step_rps(k.N, k.D), true);     // free choice: identical bits
  const int kcap = k.max_len;
  const bool use_graph = s->use_graphs && !s->profile;
  GraphKey key{k.B, k.Ls, k.N, k.D, k.max_len, 3, kcap, k.variant};
  s->graphs_current();
  auto it = s->graphs.find(key);
  if (!use_graph || (it == s->graphs.end() && !s->warmed.count(key))) {
    s->warmed.insert(key);          // first use of a shape runs eagerly once (function attributes are set outside capture)
    TTX_TRY(run_step(s, j.st, k, kcap));
    TTX_TRY(launch_accept_and_commit(s, j.st, j.g, false));
    ++j.launched;
    return TTX_OK;
  }
  if (it == s->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(j.st, hipStreamCaptureModeThreadLocal));
    int rc = run_step(s, j.st, k, kcap);
    if (rc == TTX_OK) rc = launch_accept_and_commit(s, j.st, j.g, false);
    hipError_t e = hipStreamEndCapture(j.st, &graph);
    if (rc != TTX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    if (s->graphs.size() > 512) s->drop_graphs();
    it = s->graphs.emplace(key, exec).first;
  }
  HIP_TRY(hipGraphLaunch(it->second, j.st));
  ++j.launched;
  return TTX_OK;
}

extern "C" int ttx_greedy_speculative_generate_pool(ttx_session** sessions, int n_sessions, const int64_t* d_src, int R_total,
                                                    int Ls_all, const int32_t* h_len, int capacity, const ttx_gen_params* p,
                                                    int64_t* d_out, int16_t* d_traj, int32_t* d_fin_step, ttx_gen_stats* stats,
                                                    void* stream) {
  if (!sessions || n_sessions <= 0 || !d_src || R_total < 0 || !h_len || capacity <= 0 || capacity > 4096 || !p || !d_out || !d_traj ||
      !d_fin_step)
    return fail(TTX_ERR_INVALID, "bad argument to ttx_greedy_speculative_generate_pool");
  if (R_total == 0) return TTX_OK;
  if (p->max_len > 32000) return fail(TTX_ERR_INVALID, "max_len too large for the int16 trace");
  // the longest row fixes the slot width; rows are admitted in the order given (length-sorted lists pad least: a chunk
  // is encoded at the width of its longest row)
  int len_max = 0;
  for (int i = 0; i < R_total; ++i) len_max = std::max(len_max, (int)h_len[i]);
  if (len_max > Ls_all) return fail(TTX_ERR_INVALID, "row length beyond the source matrix width");
  // slot width: 192 positions, beyond that in steps of 64 — calls whose longest sources differ share workspaces and
  // the captured step graph (a slot's keys are bounded by its own source length, so the padding costs memory only:
  // 0.8 GB of cross K/V per 512-slot pool at 192)
  int Ls_cap = std::min(std::max(192, ((len_max + 63) / 64) * 64), sessions[0]->m->cfg.max_positions);
  Ls_cap = std::max(Ls_cap, std::max(2, len_max));
  TTX_TRY(gen_validate(sessions[0], d_src, capacity, Ls_cap, p, d_out, false));
  HIP_TRY(hipSetDevice(sessions[0]->m->device));
  release_retired();
  for (int i = 0; i < n_sessions; ++i) TTX_TRY(session_alive(sessions[i]));
  EventGuard ready_guard;
  HIP_TRY(hipEventCreateWithFlags(&ready_guard.e, hipEventDisableTiming));
  hipEvent_t ready = ready_guard.e;
  HIP_TRY(hipEventRecord(ready, (hipStream_t)stream));
  bool hung = false;
  const int n_jobs = std::min(n_sessions, cdiv(R_total, std::max(1, std::min(capacity / 2, 32))));   // short lists: several small pools
  const int C = std::min(capacity, std::max(1, cdiv(R_total, n_jobs)));      // never more slots than a fair share of the rows
  std::vector<PoolJob> jobs(n_jobs);
  int rc_final = TTX_OK;
  for (int i = 0; i < n_jobs && rc_final == TTX_OK; ++i) {
    ttx_session* s = sessions[i];
    TTX_TRY(ensure_own_stream(s));
    HIP_TRY(hipStreamWaitEvent(s->own_stream, ready, 0));
    rc_final = pool_start(jobs[i], s, s->own_stream, C, Ls_cap, p, d_out, d_traj, d_fin_step);
  }
  const int min_admit = std::max(1, C / 4);                    // admissions of at least a quarter pool (or into an empty one)
  const bool serial = sessions[0]->profile;
  int cursor = 0, done = 0;
  while (done < n_jobs && rc_final == TTX_OK) {
    bool progressed = false;
    // profiling sessions (bench.py's roofline pass): the pools run ONE AFTER ANOTHER, so that the event pair around a GEMM launch
    // measures that launch and not the kernels of three other pools sharing the device with it
    int only = -1;
    if (serial)
      for (int i = n_jobs - 1; i >= 0; --i)
        if (jobs[i].phase != 3) only = i;
    for (int i = 0; i < n_jobs && rc_final == TTX_OK; ++i) {
      if (serial && i != only) continue;
      PoolJob& j = jobs[i];
      ttx_session* s = j.s;
      volatile HostInfo* hi = s->host_info;
      if (j.phase == 1) {
        if (j.launched > 0 && hi->steps_done < j.launched) {                  // the step in flight has not published yet
          // never spin forever (no HIP call in the polling loop: only the clock)
          if ((++j.idle_spins & 0xffff) == 0 && watchdog_expired(j.last_progress)) {
            rc_final = session_hung(s);
            hung = true;
            break;
          }
          continue;
        }
        j.idle_spins = 0;
        j.last_progress = std::chrono::steady_clock::now();
        int n_act = (j.launched == 0) ? 0 : hi->n_active;
        const int free_slots = C - n_act;
        // serial pass: every pool decodes an equal contiguous share of the rest of the list (what it takes at once when the list
        // fits the pools; otherwise its own continuous batching over that share, one tail per pool as in the concurrent run)
        if (serial && j.quota_end < 0) j.quota_end = cursor + cdiv(R_total - cursor, n_jobs - i);
        const int list_end = serial ? j.quota_end : R_total;
        if (cursor < list_end && (n_act == 0 || free_slots >= min_admit)) {
          // the last rows of the list are shared out over the pools still running, so that they drain together
          // instead of one pool swallowing the rest and finishing alone
          int n_running = 0;
          for (const PoolJob& o : jobs) n_running += (o.phase == 1);
          const int remaining = list_end - cursor;
          int share = std::max(1, cdiv(remaining, std::max(1, serial ? 1 : n_running)));
          // first fill of a list that fits the pools at once: an equal share for every pool not yet started (the pools
          // before this one have taken theirs), so that nothing is left waiting for a later admission
          if (!serial && j.launched == 0 && (long long)n_jobs * C >= R_total) share = std::max(1, cdiv(remaining, n_jobs - i));
          const int take = std::min({free_slots, remaining, share});
          int Ls_new = 2;                                                      // longest row of the chunk
          for (int r = cursor; r < cursor + take; ++r) Ls_new = std::max(Ls_new, (int)h_len[r]);
          rc_final = pool_admit(j, d_src + (size_t)cursor * Ls_all, Ls_all, take, Ls_new, cursor, d_out, d_traj, d_fin_step);
          if (rc_final != TTX_OK) break;
          cursor += take;
          n_act += take;
        }
        if (n_act == 0) {
          if (hipEventRecord(s->ev_c, j.st) != hipSuccess ||
              hipMemcpyAsync(s->host_state, s->state.as<DecState>(), sizeof(DecState), hipMemcpyDeviceToHost, j.st) != hipSuccess ||
              hipEventRecord(s->ev_done, j.st) != hipSuccess) {
            rc_final = fail(TTX_ERR_HIP, "slot pool: enqueueing the final state read-back failed");
            break;
          }
          j.phase = 2;
        } else {
          if (j.launched > (long long)(p->max_len + 2) * (R_total + 1)) { rc_final = fail(TTX_ERR_HIP, "decode loop failed to terminate"); break; }
          rc_final = pool_launch_step(j, n_act);
        }
        progressed = true;
      } else if (j.phase == 2) {
        if (hipEventQuery(s->ev_done) == hipSuccess) {
          const DecState& hs = *s->host_state;
          if (stats) {
            stats->model_calls += hs.steps;
            stats->accepted_tokens += hs.accepted;
            stats->produced_tokens += hs.produced;
            stats->verified_positions += hs.verified_positions;
            stats->kv_prefix_positions += hs.kv_prefix_positions;
            stats->src_positions += hs.src_positions;
            stats->src_tokens_padded += j.src_tokens_padded;
            collect_gemm_profile(s, j.st);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s->ev_a, s->ev_c) == hipSuccess) stats->decode_ms += ms;
          }
          if (hs.error == 3 && rc_final == TTX_OK)
            rc_final = fail(TTX_ERR_ROW_REPLAY, "a row emitted PAD inside its sequence: decode the batches as given");
          else if (hs.error && rc_final == TTX_OK)
            rc_final = fail(TTX_ERR_HIP, "slot pool bookkeeping failed (admitted more rows than free slots)");
          j.phase = 3;
          ++done;
          progressed = true;
        }
      }
    }
    if (!progressed) __builtin_ia32_pause();
  }
  if (hung) {
    // a stream that never published may never drain: do not wait on any of them, and retire every session of this call
    // (their workspaces may still be written by whatever is stuck)
    for (int i = 0; i < n_jobs; ++i) jobs[i].s->dead = true;
  } else {
    for (int i = 0; i < n_jobs; ++i)
      if (jobs[i].s && jobs[i].s->own_stream) (void)hipStreamSynchronize(jobs[i].s->own_stream);
  }
  if (stats) stats->status = rc_final;
  return rc_final;
}

static int generate_many_impl(ttx_session** sessions, int n_sessions, int n_batches, const int64_t* const* d_src, const int* B,
                              const int* Ls, const ttx_gen_params* p, int64_t* const* d_out, int16_t* const* d_traj,
                              int32_t* const* d_fin, ttx_gen_stats* stats, void* stream) {
  if (!sessions || n_sessions <= 0 || n_batches < 0 || !d_src || !B || !Ls || !p || !d_out)
    return fail(TTX_ERR_INVALID, "bad argument to ttx_greedy_speculative_generate_many");
  if (d_traj)
    for (int i = 0; i < n_batches; ++i)
      if (!d_traj[i] || !d_fin || !d_fin[i]) return fail(TTX_ERR_INVALID, "missing trace buffer for a batch");
  for (int i = 0; i < n_batches; ++i) TTX_TRY(gen_validate(sessions[0], d_src[i], B[i], Ls[i], p, d_out[i], false));
  if (sessions[0]->profile) n_sessions = 1;      // profiling: one batch at a time, so that an event pair times its own launch
  HIP_TRY(hipSetDevice(sessions[0]->m->device));
  release_retired();
  hipStream_t caller = (hipStream_t)stream;
  // inputs were produced on the caller's stream: every session stream waits for it once
  for (int i = 0; i < n_sessions; ++i) TTX_TRY(session_alive(sessions[i]));
  EventGuard ready_guard;
  HIP_TRY(hipEventCreateWithFlags(&ready_guard.e, hipEventDisableTiming));
  hipEvent_t ready = ready_guard.e;
  HIP_TRY(hipEventRecord(ready, caller));
  bool hung = false;
  std::vector<GenJob> jobs(n_sessions);
  for (int i = 0; i < n_sessions; ++i) {
    TTX_TRY(ensure_own_stream(sessions[i]));
    HIP_TRY(hipStreamWaitEvent(sessions[i]->own_stream, ready, 0));
  }
  const int D1 = p->draft_len + 1;
  int next = 0, done = 0, rc_final = TTX_OK;
  while (done < n_batches) {
    bool progressed = false;
    for (int i = 0; i < n_sessions; ++i) {
      GenJob& j = jobs[i];
      ttx_session* s = sessions[i];
      if (j.phase == 0) {
        if (next < n_batches) {
          j.batch = next++;
          int rc = gen_start(j, s, s->own_stream, d_src[j.batch], B[j.batch], Ls[j.batch], p, d_out[j.batch],
                             stats ? &stats[j.batch] : nullptr, false, d_traj ? d_traj[j.batch] : nullptr,
                             d_traj ? d_fin[j.batch] : nullptr);
          if (rc != TTX_OK) { rc_final = rc; done = n_batches; break; }
          progressed = true;
        }
        continue;
      }
      volatile HostInfo* hi = s->host_info;
      if (j.phase == 1) {
        if (hi->stop) {
          // stop is published by the accept kernel (or by loop init): nothing further to enqueue
          if (j.launched == 0 && hipStreamQuery(s->own_stream) != hipSuccess) continue;   // init not yet run
          int rc = gen_finish_enqueue(j);
          if (rc != TTX_OK) { rc_final = rc; done = n_batches; break; }
          progressed = true;
        } else if (hi->steps_done >= j.launched && (j.launched > 0 || hipStreamQuery(s->own_stream) == hipSuccess)) {
          if (j.launched > p->max_len + 2) { rc_final = fail(TTX_ERR_HIP, "decode loop failed to terminate"); done = n_batches; break; }
          int rc = gen_launch_step(j, hi->width + D1);
          if (rc != TTX_OK) { rc_final = rc; done = n_batches; break; }
          progressed = true;
          j.idle_spins = 0;
          j.last_progress = std::chrono::steady_clock::now();
        } else if (j.launched > 0 && (++j.idle_spins & 0xffff) == 0 && watchdog_expired(j.last_progress)) {
          rc_final = session_hung(s);                                                       // no HIP call while polling
          hung = true;
          done = n_batches;
          break;
        }
      } else if (j.phase == 2) {
        if (hipEventQuery(s->ev_done) == hipSuccess) {
          int rc = gen_finish_collect(j);
          if (rc != TTX_OK && rc_final == TTX_OK) rc_final = rc;
          ++done;
          progressed = true;
        }
      }
    }
    if (!progressed) __builtin_ia32_pause();
  }
  if (hung) {
    for (int i = 0; i < n_sessions; ++i) sessions[i]->dead = true;      // never wait for a stream that may be stuck
  } else {
    for (int i = 0; i < n_sessions; ++i) (void)hipStreamSynchronize(sessions[i]->own_stream);
  }
  return rc_final;
}

extern "C" int ttx_greedy_speculative_generate_many(ttx_session** sessions, int n_sessions, int n_batches,
                                                    const int64_t* const* d_src, const int* B, const int* Ls,
                                                    const ttx_gen_params* p, int64_t* const* d_out, ttx_gen_stats* stats,
                                                    void* stream) {
  return generate_many_impl(sessions, n_sessions, n_batches, d_src, B, Ls, p, d_out, nullptr, nullptr, stats, stream);
}

// The same engine under the per-row width rule: rows neither wait for nor are cut short by the other rows of their
// device batch, and each row's front after every step is returned, so the caller may group rows by length and still
// reproduce, batch by batch, what the reference's loop does to the batches it was given (decoding.py replays it).
extern "C" int ttx_greedy_speculative_generate_rows(ttx_session** sessions, int n_sessions, int n_batches,
                                                    const int64_t* const* d_src, const int* B, const int* Ls,
                                                    const ttx_gen_params* p, int64_t* const* d_out, int16_t* const* d_traj,
                                                    int32_t* const* d_fin_step, ttx_gen_stats* stats, void* stream) {
  if (!d_traj || !d_fin_step) return fail(TTX_ERR_INVALID, "ttx_greedy_speculative_generate_rows needs the trace buffers");
  if (p && p->max_len > 32000) return fail(TTX_ERR_INVALID, "max_len too large for the int16 trace");
  return generate_many_impl(sessions, n_sessions, n_batches, d_src, B, Ls, p, d_out, d_traj, d_fin_step, stats, stream);
}

// ------------------------------------------------------------------------------------------------
// Beam-speculative bookkeeping
extern "C" int ttx_nucleus_mask(ttx_session* s, const float* d_logits, int rows, int V, float nucleus, int n_best, float fill,
                                float* d_out, void* stream) {
  if (!s || !d_logits || !d_out || rows < 0 || V <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_nucleus_mask");
  if (V > 64 * NUC_VPL) return fail(TTX_ERR_INVALID, "ttx_nucleus_mask: vocabulary larger than 1024");
  if (n_best < 1 || n_best > NUC_MAX_KEEP) return fail(TTX_ERR_INVALID, "ttx_nucleus_mask: n_best must be in [1,32]");
  if (rows == 0) return TTX_OK;
  HIP_TRY(hipSetDevice(s->m->device));
  NucleusArgs a{d_logits, rows, V, nucleus, n_best, fill, d_out, nullptr, 0, nullptr};
  hipLaunchKernelGGL(k_nucleus, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

extern "C" int ttx_accepted_lengths(ttx_session* s, const float* d_logits, const int64_t* d_drafts, int R, int D, int V,
                                    float nucleus, int n_best, int32_t* d_n_ok, void* stream) {
  if (!s || !d_logits || !d_drafts || !d_n_ok || R < 0 || D <= 0 || V <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_accepted_lengths");
  if (V > 64 * NUC_VPL) return fail(TTX_ERR_INVALID, "ttx_accepted_lengths: vocabulary larger than 1024");
  if (n_best < 1 || n_best > NUC_MAX_KEEP) return fail(TTX_ERR_INVALID, "ttx_accepted_lengths: n_best must be in [1,32]");
  if (R == 0) return TTX_OK;
  HIP_TRY(hipSetDevice(s->m->device));
  NucleusArgs a{d_logits, R, V, nucleus, n_best, 0.f, nullptr, d_drafts, D, d_n_ok};
  hipLaunchKernelGGL(k_nucleus, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

extern "C" int ttx_ragged_topk(ttx_session* s, const float* d_score, const int32_t* d_offsets, int G, int max_group, int k,
                               float* d_top, int64_t* d_idx, void* stream) {
  if (!s || !d_score || !d_offsets || !d_top || !d_idx || G < 0 || k < 1 || max_group < 1)
    return fail(TTX_ERR_INVALID, "bad argument to ttx_ragged_topk");
  if ((size_t)max_group * 4 > 150 * 1024) return fail(TTX_ERR_INVALID, "ttx_ragged_topk: group larger than the LDS image");
  if (G == 0) return TTX_OK;
  HIP_TRY(hipSetDevice(s->m->device));
  const size_t lds = (size_t)max_group * 4;
  if (lds > 64 * 1024 && !s->attr_topk) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ragged_topk), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    s->attr_topk = true;
  }
  RaggedTopkArgs a{d_score, d_offsets, k, d_top, d_idx};
  hipLaunchKernelGGL(k_ragged_topk, dim3(G), dim3(256), lds, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Beam-search speculative decoding, whole loop native (speculative_decoding.py:428-598 all drafts, :600-845 smart drafts).
// Device kernels of one iteration: see "Native beam-speculative loop" in ttx_kernels.hip.h.  The host keeps the loop
// scalars of the reference (draft_len, width, num_of_empty_columns, postn_after_the_last_meaning_token, possible_draft_len)
// and learns {candidates holding EOS, longest new row, accepted-token marks} of every iteration from pinned words the
// last kernel of the iteration publishes — no stream synchronisation inside the loop.
struct BeamJob {
  ttx_session* s = nullptr;
  hipStream_t st = nullptr;
  ttx_beam_params p{};
  int B = 0, Ls = 0, K = 0, N = 0, D0 = 0, n_lib = 0, lib_ld = 0, max_cand = 0, gen_ld = 0, Lc = 0;
  bool smart = false;
  // loop scalars
  int n_cand = 0, beam = 1, dl = 0, width = 1, empty_cols = 0, after_last = 1, room = 0, prev_dl = 0, cur = 0;
  int launched = 0;
  int n_eos = 0;             // candidates holding EOS after the previous iteration (they are not decoded)
  int variant = GV_BIG;      // GemmVariant of the iteration about to be enqueued (from the live candidates; identical bits)
  int phase = 0;             // 0 idle, 1 iteration in flight, 2 finishing, 3 done
  bool any_iteration = false;
  ttx_beam_stats acc{};
  int64_t* d_out = nullptr;
  ttx_beam_stats* stats = nullptr;
  int rc = TTX_OK;
  std::chrono::steady_clock::time_point last_progress = std::chrono::steady_clock::now();
  unsigned idle_spins = 0;
};

static int beam_validate(const ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_params* p, const int64_t* d_out) {
  if (!s || !d_src || !p || !d_out || B <= 0 || Ls <= 1) return fail(TTX_ERR_INVALID, "bad argument to ttx_beam_speculative_generate");
  const ttx_config& c = s->m->cfg;
  if (p->n_best < 1 || p->n_best > NUC_MAX_KEEP) return fail(TTX_ERR_INVALID, "n_best must be in [1,32]");
  if (c.vocab_size > 64 * NUC_VPL) return fail(TTX_ERR_INVALID, "vocabulary larger than 1024");
  if (p->n_drafts <= 0) return fail(TTX_ERR_REFERENCE, "The number of drafts must be greater than 0");
  if (p->n_drafts > BS_MAX_SLOTS) return fail(TTX_ERR_INVALID, "n_drafts beyond the 64 draft slots of the bookkeeping kernels");
  if (p->pad_token == p->replace_token || p->eos_token == p->replace_token || p->eos_token == p->pad_token)
    return fail(TTX_ERR_REFERENCE, "pad, eos and replace tokens must be pairwise different");
  if (p->pad_token != c.pad_token) return fail(TTX_ERR_INVALID, "generator pad token differs from the model's");
  if (p->max_len < 3)          // possible_draft_len = max_len - 2 < 1: the reference's loop never runs and it returns an unbound name
    return fail(TTX_ERR_REFERENCE, "max_len < 3: the reference's loop body never runs (UnboundLocalError: new_candidates)");
  if (p->smart_drafts_mode && Ls - 5 <= 0) return fail(TTX_ERR_REFERENCE, "The number of drafts must be greater than 0");
  const int D = clamp_draft_len(p->draft_len, 5, 200);
  if (p->max_len + D + 3 > c.max_positions) return fail(TTX_ERR_INVALID, "max_len + draft_len exceeds the positional table");
  return TTX_OK;
}

static int beam_start(BeamJob& j, ttx_session* s, hipStream_t st, const int64_t* d_src, int B, int Ls, const ttx_beam_params* p,
                      int64_t* d_out, ttx_beam_stats* stats) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers, V = c.vocab_size;
  j = BeamJob{};
  j.s = s; j.st = st; j.p = *p; j.B = B; j.Ls = Ls; j.K = p->n_best; j.N = p->n_drafts; j.smart = p->smart_drafts_mode != 0;
  j.d_out = d_out; j.stats = stats;
  const int Dreq = clamp_draft_len(p->draft_len, 5, 200);            // :278-284
  // all drafts: make_drafts(src[:, 1:], draft_len, N, 5, 200) -> D0 = draft_len; smart: windows of draft_len + 1 tokens
  // (clamped again by make_drafts, drafting.py:48) whose first token is the key -> D0 = that - 1
  j.lib_ld = clamp_draft_len(Dreq + 1, 5, 200);
  j.D0 = j.smart ? j.lib_ld - 1 : Dreq;
  j.n_lib = Ls - 5;
  j.max_cand = B * j.K;
  j.gen_ld = p->max_len + j.D0 + 2;
  j.Lc = p->max_len + j.D0 + 2;
  const size_t MC = (size_t)j.max_cand;
  const size_t Mmax = MC * step_rps(j.N, j.D0);
  int rc = TTX_OK;
  auto need = [&](Buf& b, size_t bytes) { if (rc == TTX_OK) rc = ensure(b, bytes, st); };
  need(s->tok_src, (size_t)B * Ls * 4); need(s->src_valid, (size_t)B * Ls); need(s->memory, (size_t)B * Ls * d * 4);
  need(s->memkv, (size_t)B * Ls * Ld * 2 * d * 4);
  need(s->drafts, MC * j.N * std::max(j.D0, 1) * 4);
  need(s->gen, MC * j.gen_ld * 4); need(s->front, MC * 4); need(s->act_idx, MC * 4);
  need(s->pred, Mmax * 4); need(s->state, sizeof(DecState)); need(s->logits, Mmax * V * 4);
  for (int i = 0; i < 2; ++i) { need(s->tk[i], (size_t)Ld * MC * j.Lc * d * 4); need(s->tv[i], (size_t)Ld * MC * j.Lc * d * 4); }
  need(s->t_prev_len, MC * 4); need(s->t_slot_of, MC * 4); need(s->t_src_of, MC * 4);
  need(s->bs_cand_next, MC * j.gen_ld * 8); need(s->bs_len_next, MC * 4); need(s->bs_fin_next, MC); need(s->bs_logp_next, MC * 4);
  need(s->bs_len, MC * 4); need(s->bs_fin, MC); need(s->bs_active, MC); need(s->bs_logp, MC * 4); need(s->bs_per_cand, MC * 4);
  need(s->bs_best_n, MC * 4); need(s->bs_best_slot, MC * 4); need(s->bs_chosen, MC * std::max(j.D0, 1) * 8);
  need(s->bs_hit, MC * (size_t)j.N * std::max(j.D0, 1));
  need(s->bs_parent, MC * 4); need(s->bs_parent_draft, MC * 4); need(s->bs_mark, MC * 4);
  need(s->bs_drafts_src, (size_t)B * (j.smart ? (size_t)j.n_lib * j.lib_ld : (size_t)j.N * j.D0) * 4);
  need(s->bs_cnt, sizeof(BeamCounters));
  const size_t dl1 = (size_t)j.D0 + 1;
  need(s->leaf_score, MC * dl1 * j.K * 4); need(s->leaf_tok, MC * dl1 * j.K * 4); need(s->leaf_cnt, MC * dl1 * 4);
  need(s->beam_summary, 8 * 4);
  const size_t Macts = std::max(Mmax, (size_t)B * Ls);
  if (rc == TTX_OK) rc = ensure_acts(s, st, Macts, 1);
  need(s->qkv, std::max((size_t)Ld * Mmax, (size_t)B * Ls) * 3 * d * 4);
  need(s->slab, sizeof(float) * 16 * Macts * d);
  s->graphs_current();
  TTX_TRY(rc);
  if ((size_t)j.K * dl1 > 1023 || 2 * (size_t)j.K * dl1 * j.K * 4 > 150 * 1024)
    return fail(TTX_ERR_INVALID, "too many leaves per source for the selection kernel's LDS image");
  if (!s->beam_host) {
    if (hipHostMalloc((void**)&s->beam_host, sizeof(BeamHost), hipHostMallocMapped) != hipSuccess)
      return fail(TTX_ERR_NOMEM, "hipHostMalloc failed");
  }
  std::memset(s->beam_host, 0, sizeof(BeamHost));

  s->ev_used = 0;
  HIP_TRY(hipEventRecord(s->ev_a, st));
  // encoder + cross K/V once per source (:439 / :626); drafts (:430) or the draft library (:603-615)
  TTX_TRY(prepare_tokens(st, d_src, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B * Ls, c.pad_token));
  TTX_TRY(run_encoder(s, st, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B, Ls, s->memory.as<float>()));
  const int gv = variant_for_rows(s, (long long)B * Ls, false);
  TTX_TRY(launch_gemm(s, st, s->memory.as<float>(), d, m->p(m->cross_kv_w), d, m->p(m->cross_kv_b), s->memkv.as<float>(),
                      Ld * 2 * d, nullptr, B * Ls, Ld * 2 * d, d, false, 0, 0, gv));
  if (j.smart)
    TTX_TRY(launch_make_drafts<int>(st, s->tok_src.as<int>(), Ls, 0, B, Ls, j.n_lib, j.lib_ld, p->eos_token, p->pad_token,
                                    p->replace_token, s->bs_drafts_src.as<int>()));
  else
    TTX_TRY(launch_make_drafts<int>(st, s->tok_src.as<int>(), Ls, 1, B, Ls - 1, j.N, j.D0, p->eos_token, p->pad_token,
                                    p->replace_token, s->bs_drafts_src.as<int>()));
  hipLaunchKernelGGL(k_bs_init, dim3(64), dim3(256), 0, st, s->bs_cand_next.as<int64_t>(), j.gen_ld, s->bs_len_next.as<int>(),
                     s->bs_fin_next.as<uint8_t>(), s->bs_logp_next.as<float>(), s->bs_parent.as<int>(), s->bs_parent_draft.as<int>(),
                     j.max_cand, B, p->bos_token, p->pad_token, s->bs_cnt.as<BeamCounters>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev_b, st));
  // candidate c of an iteration belongs to source c / beam: the tree kernels read the map from t_src_of, filled per iteration
  j.n_cand = B; j.beam = 1; j.dl = j.D0; j.width = 1; j.empty_cols = 0; j.after_last = 1;
  j.room = p->max_len - j.after_last - 1;
  j.cur = 0; j.prev_dl = j.D0;
  j.phase = 1;
  return TTX_OK;
}

__global__ void k_bs_src_of(int* src_of, int n, int beam) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) src_of[i] = i / beam;
}

// The kernels of one iteration, in order, as a function of the job's scalars alone (so that the sequence can be captured
// once per shape and replayed).  `first`: no cache to derive (every candidate is a fresh <BOS> row); `cur`: which of the
// two cache buffers holds the parents' caches.
static int beam_enqueue_iter(const BeamJob& j, bool first, int cur) {
  ttx_session* s = j.s;
  hipStream_t st = j.st;
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers, V = c.vocab_size;
  const int dl = j.dl, MC = j.max_cand;
  const long long cache_seq = (long long)j.Lc * d, cache_layer = (long long)MC * cache_seq;
  BeamPrepArgs pa{};
  pa.cand_next = s->bs_cand_next.as<int64_t>(); pa.ld = j.gen_ld; pa.len_next = s->bs_len_next.as<int>();
  pa.fin_next = s->bs_fin_next.as<uint8_t>(); pa.logp_next = s->bs_logp_next.as<float>();
  pa.n_cand = j.n_cand; pa.beam = j.beam; pa.dl = dl; pa.N = j.N; pa.pad = j.p.pad_token;
  pa.smart = j.smart ? 1 : 0; pa.n_lib = j.n_lib; pa.lib_ld = j.lib_ld;
  pa.drafts_all = s->bs_drafts_src.as<int>(); pa.D0 = j.D0; pa.lib = s->bs_drafts_src.as<int>();
  pa.gen = s->gen.as<int>(); pa.front = s->front.as<int>(); pa.len = s->bs_len.as<int>(); pa.active = s->bs_active.as<uint8_t>();
  pa.finished = s->bs_fin.as<uint8_t>(); pa.logp = s->bs_logp.as<float>(); pa.per_cand = s->bs_per_cand.as<int>();
  pa.drafts32 = s->drafts.as<int>();
  hipLaunchKernelGGL(k_bs_prep, dim3(MC), dim3(256), 0, st, pa);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_bs_src_of, dim3(cdiv(MC, 256)), dim3(256), 0, st, s->t_src_of.as<int>(), MC, j.beam);
  HIP_TRY(hipGetLastError());
  const int nxt = cur ^ 1;
  if (!first) {
    TreeCacheArgs ca{};
    ca.len = s->bs_len.as<int>(); ca.parent = s->bs_parent.as<int>(); ca.parent_draft = s->bs_parent_draft.as<int>();
    ca.prev_len = s->t_prev_len.as<int>(); ca.active = s->bs_active.as<uint8_t>();
    ca.k_old = s->tk[cur].as<float>(); ca.v_old = s->tv[cur].as<float>();
    ca.k_new = s->tk[nxt].as<float>(); ca.v_new = s->tv[nxt].as<float>();
    ca.cache_layer_stride = cache_layer; ca.cache_seq_stride = cache_seq;
    ca.qkv_prev = s->qkv.as<float>();
    ca.qkv_layer_stride = (long long)MC * step_rps(j.N, j.prev_dl) * 3 * d;
    ca.prev_slot_of = s->t_slot_of.as<int>(); ca.prev_N = j.N; ca.prev_D = j.prev_dl; ca.d = d;
    hipLaunchKernelGGL(k_tree_cache, dim3(MC, Ld), dim3(256), 0, st, ca);
    HIP_TRY(hipGetLastError());
  }
  BeamListArgs la{};
  la.active = s->bs_active.as<uint8_t>(); la.per_cand = s->bs_per_cand.as<int>(); la.len = s->bs_len.as<int>();
  la.n_cand = j.n_cand; la.N = j.N; la.dl = dl;
  la.act_idx = s->act_idx.as<int>(); la.slot_of = s->t_slot_of.as<int>(); la.prev_len = s->t_prev_len.as<int>();
  la.st = s->state.as<DecState>(); la.cnt = s->bs_cnt.as<BeamCounters>(); la.summary = s->beam_summary.as<int>();
  hipLaunchKernelGGL(k_bs_list, dim3(1), dim3(256), 0, st, la);
  HIP_TRY(hipGetLastError());
  // the verify step: D+1 new positions per (running candidate, draft slot) on the candidate's KV cache
  StepCtx k{};
  k.B = MC; k.Ls = j.Ls; k.N = j.N; k.D = dl; k.Lc = j.Lc; k.gen_ld = j.gen_ld; k.max_len = j.p.max_len;
  k.kcache = s->tk[nxt].as<float>(); k.vcache = s->tv[nxt].as<float>(); k.src_of = s->t_src_of.as<int>(); k.want_argmax = false;
  k.variant = j.variant;
  TTX_TRY(run_step(s, st, k, std::min(j.p.max_len, ((j.width + 63) / 64) * 64)));
  BeamHitsArgs ha{};
  ha.logits = s->logits.as<float>(); ha.V = V; ha.finished = s->bs_fin.as<uint8_t>(); ha.slot_of = s->t_slot_of.as<int>();
  ha.per_cand = s->bs_per_cand.as<int>(); ha.drafts32 = s->drafts.as<int>();
  ha.n_cand = j.n_cand; ha.N = j.N; ha.dl = dl; ha.K = j.K; ha.nucleus = 0.9975f; ha.hit = s->bs_hit.as<uint8_t>();
  BeamLeaves2Args le{};
  le.logits = s->logits.as<float>(); le.V = V; le.finished = s->bs_fin.as<uint8_t>(); le.slot_of = s->t_slot_of.as<int>();
  le.per_cand = s->bs_per_cand.as<int>(); le.drafts32 = s->drafts.as<int>(); le.logp = s->bs_logp.as<float>();
  le.hit = s->bs_hit.as<uint8_t>(); le.cnt = s->bs_cnt.as<BeamCounters>();
  le.n_cand = j.n_cand; le.N = j.N; le.dl = dl; le.K = j.K; le.bos = j.p.bos_token; le.pad = j.p.pad_token; le.smart = j.smart ? 1 : 0;
  le.best_n = s->bs_best_n.as<int>(); le.best_slot = s->bs_best_slot.as<int>(); le.chosen = s->bs_chosen.as<int64_t>();
  le.leaf_score = s->leaf_score.as<float>(); le.leaf_tok = s->leaf_tok.as<int>(); le.leaf_cnt = s->leaf_cnt.as<int>();
  const dim3 hits_grid(MC, cdiv(std::max(j.N * dl, 1), BS_HITS_WAVES));
  const size_t leaves_lds = (size_t)2 * (dl + 1) * 4;
  // logits per lane the selection keeps in registers: the smallest of 4 / 8 / 16 that covers the vocabulary (same results)
  if (V <= 256) {
    if (dl > 0) hipLaunchKernelGGL(k_bs_hits<4>, hits_grid, dim3(BS_HITS_WAVES * 64), 0, st, ha);
    hipLaunchKernelGGL(k_bs_leaves<4>, dim3(MC), dim3(BS_LEAVES_THREADS), leaves_lds, st, le);
  } else if (V <= 512) {
    if (dl > 0) hipLaunchKernelGGL(k_bs_hits<8>, hits_grid, dim3(BS_HITS_WAVES * 64), 0, st, ha);
    hipLaunchKernelGGL(k_bs_leaves<8>, dim3(MC), dim3(BS_LEAVES_THREADS), leaves_lds, st, le);
  } else {
    if (dl > 0) hipLaunchKernelGGL(k_bs_hits<NUC_VPL>, hits_grid, dim3(BS_HITS_WAVES * 64), 0, st, ha);
    hipLaunchKernelGGL(k_bs_leaves<NUC_VPL>, dim3(MC), dim3(BS_LEAVES_THREADS), leaves_lds, st, le);
  }
  HIP_TRY(hipGetLastError());
  BeamSelectArgs<int> sa{s->leaf_score.as<float>(), s->leaf_tok.as<int>(), s->leaf_cnt.as<int>(), s->gen.as<int>(), j.gen_ld, j.gen_ld,
                         j.gen_ld, s->bs_len.as<int>(), s->bs_chosen.as<int64_t>(), s->bs_best_slot.as<int>(), s->bs_fin.as<uint8_t>(),
                         j.B, j.beam, dl, j.K, j.p.pad_token, j.p.eos_token, s->bs_cand_next.as<int64_t>(), s->bs_logp_next.as<float>(),
                         s->bs_parent.as<int>(), s->bs_parent_draft.as<int>(), s->bs_mark.as<int>(), s->beam_summary.as<int>(),
                         s->bs_len_next.as<int>(), s->bs_fin_next.as<uint8_t>()};
  const size_t lds = 2 * (size_t)j.beam * (dl + 1) * j.K * 4;
  if (lds > 64 * 1024 && !s->attr_select) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_beam_select<int>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    s->attr_select = true;
  }
  hipLaunchKernelGGL(k_beam_select<int>, dim3(j.B), dim3(256), lds, st, sa);
  HIP_TRY(hipGetLastError());
  BeamHost* dev_host = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dev_host, (void*)s->beam_host, 0));
  hipLaunchKernelGGL(k_bs_publish, dim3(1), dim3(64), 0, st, s->beam_summary.as<int>(), dev_host, s->bs_cnt.as<BeamCounters>());
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

// Enqueue one iteration (the loop condition of :464 / :652 was checked by the caller): replay the captured graph of this
// shape, capturing it on its second use (the first use runs eagerly: function attributes are set outside capture).
static int beam_launch_iter(BeamJob& j) {
  ttx_session* s = j.s;
  j.dl = std::min(j.room, j.dl);                                   // :476
  const int grow = j.dl + 1 - j.empty_cols;
  if (grow > 0) j.width += grow;
  const bool first = j.launched == 0;
  const int cur = j.cur;
  const int kcap = std::min(j.p.max_len, ((j.width + 63) / 64) * 64);
  j.variant = variant_for_rows(s, (long long)std::max(1, j.n_cand - j.n_eos) * step_rps(j.N, j.dl), true);
  int rc = TTX_OK;
  if (!s->use_graphs || s->profile) {
    rc = beam_enqueue_iter(j, first, cur);
  } else {
    s->graphs_current();
    const std::vector<int> key{j.B, j.Ls, j.K, j.N, j.D0, j.smart ? 1 : 0, j.p.max_len, j.n_cand, j.beam, j.dl, j.prev_dl, cur, kcap,
                               first ? 1 : 0, j.p.pad_token, j.p.bos_token, j.p.eos_token, j.variant};
    auto it = s->beam_graphs.find(key);
    if (it == s->beam_graphs.end() && !s->beam_warmed.count(key)) {
      s->beam_warmed.insert(key);
      rc = beam_enqueue_iter(j, first, cur);
    } else {
      if (it == s->beam_graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        const auto cap0 = std::chrono::steady_clock::now();
        HIP_TRY(hipStreamBeginCapture(j.st, hipStreamCaptureModeThreadLocal));
        rc = beam_enqueue_iter(j, first, cur);
        hipError_t e = hipStreamEndCapture(j.st, &graph);
        if (rc != TTX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        if (s->beam_graphs.size() > 256) s->drop_graphs();
        it = s->beam_graphs.emplace(key, exec).first;
        s->host_captures += 1;
        s->host_capture_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - cap0).count();
      }
      HIP_TRY(hipGraphLaunch(it->second, j.st));
    }
  }
  TTX_TRY(rc);
  ++j.launched;
  j.cur = cur ^ 1;
  j.prev_dl = j.dl;
  j.any_iteration = true;
  j.last_progress = std::chrono::steady_clock::now();
  j.idle_spins = 0;
  return TTX_OK;
}

// The published summary of the iteration just finished -> the reference's loop scalars (:578-598).  Returns true when the
// loop goes on.
static bool beam_after_iter(BeamJob& j) {
  const volatile int* sm = j.s->beam_host->summary;
  const int n_eos = sm[0], min_pad = sm[1], acc_sum = sm[2], acc_cnt = sm[3], err = sm[4];
  if (err) {
    j.rc = fail(TTX_ERR_REFERENCE, "a source has fewer candidate leaves than n_best (the reference asserts here, speculative_decoding.py:195)");
    return false;
  }
  j.acc.accepted_tokens += acc_sum;
  j.acc.produced_non_pad_tokens += acc_sum + acc_cnt;
  j.n_cand = j.B * j.K;
  j.beam = j.K;
  j.n_eos = n_eos;
  if (n_eos == j.B * j.K) return false;                             // :586
  const int max_real = j.gen_ld - min_pad;                          // longest new row
  j.empty_cols = j.width - max_real;                                // min over rows of their PAD count (:592)
  j.after_last = j.width - j.empty_cols;
  j.room = j.p.max_len - j.after_last - 1;
  if (!(j.room >= 1 && j.after_last <= j.p.max_len)) return false;  // :464
  if (j.p.max_steps > 0 && j.launched >= j.p.max_steps) {
    j.rc = fail(TTX_ERR_MAX_STEPS, "beam-speculative loop exceeded max_steps (non-terminating input)");
    return false;
  }
  return true;
}

static int beam_finish_enqueue(BeamJob& j) {
  ttx_session* s = j.s;
  // new_candidates.reshape(b_size, n_best, -1): rows of the last selection, `width` columns
  if (j.width > j.p.max_len) return fail(TTX_ERR_HIP, "beam-speculative loop: result wider than max_len (internal error)");
  HIP_TRY(hipMemcpy2DAsync(j.d_out, (size_t)j.p.max_len * 8, s->bs_cand_next.p, (size_t)j.gen_ld * 8, (size_t)j.width * 8,
                           (size_t)j.B * j.K, hipMemcpyDeviceToDevice, j.st));
  HIP_TRY(hipEventRecord(s->ev_c, j.st));
  HIP_TRY(hipMemcpyAsync(s->host_state, s->bs_cnt.p, sizeof(BeamCounters), hipMemcpyDeviceToHost, j.st));
  HIP_TRY(hipEventRecord(s->ev_done, j.st));
  j.phase = 2;
  return TTX_OK;
}

static void beam_collect(BeamJob& j) {
  const BeamCounters* cn = reinterpret_cast<const BeamCounters*>(j.s->host_state);
  j.acc.model_calls = cn->model_calls;
  j.acc.input_lines = cn->input_lines;
  j.acc.running_rows = cn->running_rows;
  j.acc.verified_positions = cn->verified_positions;
  j.acc.executed_positions = cn->executed_positions;
  j.acc.kv_prefix_positions = cn->kv_prefix_positions;
  j.acc.running_candidates = cn->running_cands;
  j.acc.src_tokens_padded = (int64_t)j.B * j.Ls;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, j.s->ev_a, j.s->ev_b) == hipSuccess) j.acc.encode_ms = ms;
  if (hipEventElapsedTime(&ms, j.s->ev_b, j.s->ev_c) == hipSuccess) j.acc.decode_ms = ms;
  collect_gemm_profile(j.s, j.st);
  j.acc.out_width = j.width;
  j.acc.status = j.rc;
  if (j.stats) *j.stats = j.acc;
  j.phase = 3;
}

static_assert(sizeof(BeamCounters) <= sizeof(DecState), "the pinned read-back buffer is sized for DecState");

extern "C" int ttx_beam_speculative_generate_many(ttx_session** sessions, int n_sessions, int n_batches, const int64_t* const* d_src,
                                                  const int* B, const int* Ls, const ttx_beam_params* p, int64_t* const* d_out,
                                                  ttx_beam_stats* stats, void* stream) {
  if (!sessions || n_sessions <= 0 || n_batches < 0 || !d_src || !B || !Ls || !p || !d_out || !stats)
    return fail(TTX_ERR_INVALID, "bad argument to ttx_beam_speculative_generate_many");
  for (int i = 0; i < n_sessions; ++i) { if (!sessions[i]) return fail(TTX_ERR_INVALID, "null session"); TTX_TRY(session_alive(sessions[i])); }
  for (int i = 0; i < n_batches; ++i) TTX_TRY(beam_validate(sessions[0], d_src[i], B[i], Ls[i], p, d_out[i]));
  if (n_batches == 0) return TTX_OK;
  HIP_TRY(hipSetDevice(sessions[0]->m->device));
  release_retired();
  EventGuard ready;
  HIP_TRY(hipEventCreateWithFlags(&ready.e, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ready.e, (hipStream_t)stream));
  if (sessions[0]->profile) n_sessions = 1;      // profiling: one batch at a time (see the greedy driver)
  const int n_jobs = std::min(n_sessions, n_batches);
  for (int i = 0; i < n_jobs; ++i) {
    TTX_TRY(ensure_own_stream(sessions[i]));
    HIP_TRY(hipStreamWaitEvent(sessions[i]->own_stream, ready.e, 0));
  }
  std::vector<BeamJob> jobs(n_jobs);
  std::vector<int> batch_of(n_jobs, -1);
  int next = 0, done = 0, rc_final = TTX_OK;
  bool hung = false;
  auto fail_all = [&](int rc) { if (rc_final == TTX_OK) rc_final = rc; done = n_batches; };
  while (done < n_batches) {
    bool progressed = false;
    for (int i = 0; i < n_jobs && done < n_batches; ++i) {
      BeamJob& j = jobs[i];
      ttx_session* s = sessions[i];
      if (j.phase == 0 || j.phase == 3) {
        if (next >= n_batches) continue;
        const int b = next++;
        batch_of[i] = b;
        int rc = beam_start(j, s, s->own_stream, d_src[b], B[b], Ls[b], p, d_out[b], &stats[b]);
        if (rc == TTX_OK) rc = beam_launch_iter(j);                 // max_len >= 3: the first iteration always runs
        if (rc != TTX_OK) { fail_all(rc); break; }
        progressed = true;
      } else if (j.phase == 1) {
        if (s->beam_host->steps_done < j.launched) {               // the iteration in flight has not published yet
          if ((++j.idle_spins & 0xffff) == 0 && watchdog_expired(j.last_progress)) { hung = true; fail_all(session_hung(s)); break; }
          continue;
        }
        int rc = TTX_OK;
        if (beam_after_iter(j)) rc = beam_launch_iter(j);
        else rc = beam_finish_enqueue(j);
        if (rc != TTX_OK) { fail_all(rc); break; }
        progressed = true;
      } else if (j.phase == 2) {
        if (hipEventQuery(s->ev_done) == hipSuccess) {
          beam_collect(j);
          if (j.rc != TTX_OK && rc_final == TTX_OK) rc_final = j.rc;
          ++done;
          progressed = true;
        }
      }
    }
    if (!progressed) __builtin_ia32_pause();
  }
  if (hung) {
    for (int i = 0; i < n_jobs; ++i) sessions[i]->dead = true;
  } else {
    for (int i = 0; i < n_jobs; ++i) (void)hipStreamSynchronize(sessions[i]->own_stream);
  }
  return rc_final;
}

extern "C" int ttx_beam_speculative_generate(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_params* p,
                                             int64_t* d_out, ttx_beam_stats* stats, void* stream) {
  ttx_beam_stats local{};
  const int64_t* srcs[1] = {d_src};
  int64_t* outs[1] = {d_out};
  ttx_session* ss[1] = {s};
  if (!s) return fail(TTX_ERR_INVALID, "null session");
  const int rc = ttx_beam_speculative_generate_many(ss, 1, 1, srcs, &B, &Ls, p, outs, stats ? stats : &local, stream);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Beam-speculative batch pool (kernels and the argument for exactness: "Beam-speculative BATCH POOL" in
// ttx_loop_kernels.hip.h).  Host side: one job per session, each a pool of C source slots fed with whole batches from the
// caller's work list; an iteration is one fixed launch sequence (captured once per cache parity and GEMM variant) over every
// live candidate of every batch in the pool.
struct BeamPoolJob {
  ttx_session* s = nullptr;
  hipStream_t st = nullptr;
  ttx_beam_params p{};
  int C = 0, K = 0, N = 0, D0 = 0, lib_ld = 0, Ls_cap = 0, gen_ld = 0, Lc = 0, MC = 0;
  bool smart = false;
  BeamPoolArgs a{};
  BeamPoolIo io{};
  int launched = 0, cur = 0;
  int phase = 0;                    // 0 not started, 1 running, 2 finishing, 3 done
  int admitted_since = 0;           // sources admitted since the last published iteration
  int quota_b = -1;                 // serial (profiling) pass: end (batch index) of this pool's share of the work list
  long long src_tokens_padded = 0, admitted_rows = 0;
  unsigned idle_spins = 0;
  std::chrono::steady_clock::time_point last_progress = std::chrono::steady_clock::now();
};

static int bpool_start(BeamPoolJob& j, ttx_session* s, hipStream_t st, int C, int Ls_cap, const ttx_beam_params* p, const BeamPoolIo& io_host,
                       const int32_t* h_len, const int32_t* h_batch_of, const int32_t* h_given, int R_total, int n_batches) {
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers, V = c.vocab_size;
  j = BeamPoolJob{};
  j.s = s; j.st = st; j.p = *p; j.C = C; j.K = p->n_best; j.N = p->n_drafts; j.smart = p->smart_drafts_mode != 0; j.Ls_cap = Ls_cap;
  const int Dreq = clamp_draft_len(p->draft_len, 5, 200);
  j.lib_ld = clamp_draft_len(Dreq + 1, 5, 200);
  j.D0 = j.smart ? j.lib_ld - 1 : Dreq;
  j.gen_ld = p->max_len + j.D0 + 2;
  j.Lc = p->max_len + j.D0 + 2;
  j.MC = C * j.K;
  const size_t MC = (size_t)j.MC;
  const size_t Mmax = MC * step_rps(j.N, j.D0);
  const size_t kv_row = (size_t)Ld * 2 * d;
  int rc = TTX_OK;
  auto need = [&](Buf& b, size_t bytes) { if (rc == TTX_OK) rc = ensure(b, bytes, st); };
  need(s->tok_src, (size_t)C * Ls_cap * 4); need(s->valid_new, (size_t)C * Ls_cap); need(s->src_valid, (size_t)C * Ls_cap);
  need(s->memory, (size_t)C * Ls_cap * d * 4); need(s->memkv_new, (size_t)C * Ls_cap * kv_row * 4); need(s->memkv, (size_t)C * Ls_cap * kv_row * 4);
  need(s->bp_tok, (size_t)C * Ls_cap * 4);
  need(s->bp_enc_qkv, (size_t)C * Ls_cap * 3 * d * 4);      // admissions must not touch the step's Q/K/V rows (see run_encoder)
  need(s->drafts_new, (size_t)C * j.N * j.D0 * 4); need(s->bs_drafts_src, (size_t)C * j.N * j.D0 * 4);
  need(s->drafts, MC * j.N * j.D0 * 4);
  need(s->gen, MC * j.gen_ld * 4); need(s->front, MC * 4); need(s->act_idx, MC * 4);
  need(s->pred, Mmax * 4); need(s->state, sizeof(DecState)); need(s->logits, Mmax * V * 4);
  need(s->tk[0], (size_t)Ld * MC * j.Lc * d * 4); need(s->tv[0], (size_t)Ld * MC * j.Lc * d * 4);      // one buffer: candidate -> slot map
  need(s->t_prev_len, MC * 4); need(s->t_slot_of, MC * 4); need(s->t_src_of, MC * 4); need(s->bp_cand_len, MC * 4);
  need(s->bs_cand_next, MC * j.gen_ld * 8); need(s->bs_len_next, MC * 4); need(s->bs_fin_next, MC); need(s->bs_logp_next, MC * 4);
  need(s->bs_len, MC * 4); need(s->bs_fin, MC); need(s->bs_active, MC); need(s->bs_logp, MC * 4); need(s->bs_per_cand, MC * 4);
  need(s->bs_best_n, MC * 4); need(s->bs_best_slot, MC * 4); need(s->bs_chosen, MC * j.D0 * 8);
  need(s->bs_hit, MC * (size_t)j.N * j.D0); need(s->bs_mark, MC);            // bs_mark: the live flags of the pool
  need(s->bs_parent, MC * 4); need(s->bs_parent_draft, MC * 4); need(s->bp_cand, MC * 16);     // cand_dl, cand_batch, cache_slot, cache_slot_parent
  need(s->bs_cnt, sizeof(BeamCounters));
  const size_t dl1 = (size_t)j.D0 + 1;
  need(s->leaf_score, MC * dl1 * j.K * 4); need(s->leaf_tok, MC * dl1 * j.K * 4); need(s->leaf_cnt, MC * dl1 * 4);
  need(s->beam_summary, 8 * 4); need(s->bp_grp, 4 * 4);
  need(s->bp_row_of, (size_t)C * 4 * 3);                        // row_of, slot_batch, src_state
  need(s->bp_batch, (size_t)C * 4 * 10);                        // ten per-batch-slot arrays
  need(s->bp_src_acc, (size_t)C * 32);
  need(s->bp_new_slot, (size_t)C * 4);
  need(s->bp_io, sizeof(BeamPoolIo) + ((size_t)R_total * 2 + n_batches) * 4);
  const size_t Macts = std::max(Mmax, (size_t)C * Ls_cap);
  if (rc == TTX_OK) rc = ensure_acts(s, st, Macts, 1);
  need(s->qkv, std::max((size_t)Ld * Mmax, (size_t)C * Ls_cap) * 3 * d * 4);
  need(s->slab, sizeof(float) * 16 * Macts * d);
  s->graphs_current();
  TTX_TRY(rc);
  if ((size_t)j.K * dl1 > 1023 || 2 * (size_t)j.K * dl1 * j.K * 4 > 150 * 1024)
    return fail(TTX_ERR_INVALID, "too many leaves per source for the selection kernel's LDS image");
  if (!s->bp_host) {
    if (hipHostMalloc((void**)&s->bp_host, sizeof(BeamPoolHost), hipHostMallocMapped) != hipSuccess)
      return fail(TTX_ERR_NOMEM, "hipHostMalloc failed");
  }
  std::memset(s->bp_host, 0, sizeof(BeamPoolHost));
  BeamPoolHost* dev_host = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dev_host, (void*)s->bp_host, 0));
  s->ev_used = 0;
  HIP_TRY(hipEventRecord(s->ev_a, st));
  // the caller-side pointer block and the work list's lengths live in one device allocation:
  // [BeamPoolIo][len_all: R_total][batch_all: R_total][given_all: n_batches]
  char* io_dev = s->bp_io.as<char>();
  int* len_dev = reinterpret_cast<int*>(io_dev + sizeof(BeamPoolIo));
  int* batch_dev = len_dev + R_total;
  int* given_dev = batch_dev + R_total;
  j.io = io_host;
  j.io.len_all = len_dev; j.io.batch_all = batch_dev; j.io.given_all = given_dev;
  HIP_TRY(hipMemcpyAsync(io_dev, &j.io, sizeof(BeamPoolIo), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(len_dev, h_len, (size_t)R_total * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(batch_dev, h_batch_of, (size_t)R_total * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(given_dev, h_given, (size_t)n_batches * 4, hipMemcpyHostToDevice, st));
  BeamPoolArgs& a = j.a;
  a.C = C; a.K = j.K; a.N = j.N; a.D0 = j.D0; a.Ls_cap = Ls_cap; a.max_len = p->max_len; a.ld = j.gen_ld;
  a.smart = j.smart ? 1 : 0; a.lib_ld = j.lib_ld; a.pad = p->pad_token; a.bos = p->bos_token; a.eos = p->eos_token; a.repl = p->replace_token;
  a.max_steps = p->max_steps;
  a.row_of = s->bp_row_of.as<int>(); a.slot_batch = a.row_of + C; a.src_state = a.row_of + 2 * C; a.src_acc = s->bp_src_acc.as<int>();
  {
    int* bb = s->bp_batch.as<int>();
    a.bat_id = bb; a.bat_iter = bb + C; a.bat_dl = bb + 2 * C; a.bat_grp = bb + 3 * C; a.bat_live = bb + 4 * C; a.bat_nfin = bb + 5 * C;
    a.bat_longest_fin = bb + 6 * C; a.bat_longest_cur = bb + 7 * C; a.bat_state = bb + 8 * C; a.bat_given = bb + 9 * C;
  }
  a.cand_dl = s->bp_cand.as<int>(); a.cand_batch = a.cand_dl + j.MC; a.cache_slot = a.cand_dl + 2 * j.MC; a.cache_slot_parent = a.cand_dl + 3 * j.MC;
  a.tok = s->bp_tok.as<int>(); a.drafts_all = s->bs_drafts_src.as<int>();
  a.cand_next = s->bs_cand_next.as<int64_t>(); a.len_next = s->bs_len_next.as<int>(); a.fin_next = s->bs_fin_next.as<uint8_t>();
  a.logp_next = s->bs_logp_next.as<float>(); a.parent = s->bs_parent.as<int>(); a.parent_draft = s->bs_parent_draft.as<int>();
  a.gen = s->gen.as<int>(); a.front = s->front.as<int>(); a.len = s->bs_len.as<int>(); a.active = s->bs_active.as<uint8_t>();
  a.finished = s->bs_fin.as<uint8_t>(); a.live = s->bs_mark.as<uint8_t>(); a.logp = s->bs_logp.as<float>(); a.per_cand = s->bs_per_cand.as<int>();
  a.drafts32 = s->drafts.as<int>();
  a.chosen_slot = s->bs_best_slot.as<int>(); a.chosen = s->bs_chosen.as<int64_t>();
  a.leaf_score = s->leaf_score.as<float>(); a.leaf_tok = s->leaf_tok.as<int>(); a.leaf_cnt = s->leaf_cnt.as<int>();
  a.cnt = s->bs_cnt.as<BeamCounters>(); a.io = reinterpret_cast<const BeamPoolIo*>(io_dev); a.host = dev_host; a.dev_summary = s->bp_grp.as<int>();
  hipLaunchKernelGGL(k_bsp_init, dim3(64), dim3(256), 0, st, a, s->t_src_of.as<int>(), s->bp_cand_len.as<int>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev_b, st));
  j.phase = 1;
  return TTX_OK;
}

// Encode R new sources (rows first_row .. of the caller's matrix, Ls_new columns of it) and hand them free source slots.
static int bpool_admit(BeamPoolJob& j, const int64_t* d_src_rows, int ld_src, int R, int Ls_new, int first_row) {
  ttx_session* s = j.s;
  hipStream_t st = j.st;
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers;
  const int kv_row = Ld * 2 * d;
  hipLaunchKernelGGL(k_prepare_tokens_2d, dim3(cdiv(R * Ls_new, 256)), dim3(256), 0, st, d_src_rows, ld_src, s->tok_src.as<int>(),
                     s->valid_new.as<uint8_t>(), R, Ls_new, c.pad_token);
  HIP_TRY(hipGetLastError());
  TTX_TRY(run_encoder(s, st, s->tok_src.as<int>(), s->valid_new.as<uint8_t>(), R, Ls_new, s->memory.as<float>(), s->bp_enc_qkv.as<float>()));
  const int gv = variant_for_rows(s, (long long)R * Ls_new, false);
  TTX_TRY(launch_gemm(s, st, s->memory.as<float>(), d, m->p(m->cross_kv_w), d, m->p(m->cross_kv_b), s->memkv_new.as<float>(),
                      kv_row, nullptr, R * Ls_new, kv_row, d, false, 0, 0, gv));
  if (!j.smart)      // make_drafts(src[:, 1:], draft_len, N, 5, 200) (:430): independent of how far the row is padded
    TTX_TRY(launch_make_drafts<int>(st, s->tok_src.as<int>(), Ls_new, 1, R, Ls_new - 1, j.N, j.D0, j.p.eos_token, j.p.pad_token,
                                    j.p.replace_token, s->drafts_new.as<int>()));
  hipLaunchKernelGGL(k_bsp_admit, dim3(1), dim3(1), 0, st, j.a, s->bp_new_slot.as<int>(), s->bp_cand_len.as<int>(), R, first_row);
  HIP_TRY(hipGetLastError());
  BeamPoolFillArgs f{};
  f.new_slot = s->bp_new_slot.as<int>(); f.R = R; f.first_row = first_row;
  f.C = j.C; f.K = j.K; f.N = j.N; f.D0 = j.D0; f.ld = j.gen_ld; f.Ls_cap = j.Ls_cap; f.Ls_new = Ls_new; f.max_len = j.p.max_len;
  f.pad = j.p.pad_token; f.bos = j.p.bos_token;
  f.cand_next = j.a.cand_next; f.len_next = j.a.len_next; f.fin_next = j.a.fin_next; f.logp_next = j.a.logp_next; f.parent = j.a.parent;
  f.parent_draft = j.a.parent_draft;
  f.tok = s->bp_tok.as<int>(); f.tok_new = s->tok_src.as<int>(); f.src_valid = s->src_valid.as<uint8_t>(); f.valid_new = s->valid_new.as<uint8_t>();
  f.drafts_all = s->bs_drafts_src.as<int>(); f.drafts_new = j.smart ? nullptr : s->drafts_new.as<int>();
  f.memkv = s->memkv.as<float>(); f.memkv_new = s->memkv_new.as<float>(); f.kv_row = kv_row; f.io = j.a.io;
  hipLaunchKernelGGL(k_bsp_fill, dim3(R, 1 + Ls_new), dim3(256), 0, st, f);
  HIP_TRY(hipGetLastError());
  j.admitted_rows += R;
  j.admitted_since += R;
  j.src_tokens_padded += (long long)R * Ls_new;
  return TTX_OK;
}

static int bpool_enqueue_iter(const BeamPoolJob& j, int cur, int variant) {
  ttx_session* s = j.s;
  hipStream_t st = j.st;
  const ttx_config& c = s->m->cfg;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers, V = c.vocab_size;
  const int dl = j.D0, MC = j.MC;
  const long long cache_seq = (long long)j.Lc * d, cache_layer = (long long)MC * cache_seq;
  hipLaunchKernelGGL(k_bsp_prep, dim3(MC), dim3(256), 0, st, j.a);
  HIP_TRY(hipGetLastError());
  const int nxt = cur ^ 1;
  (void)nxt;
  TreeCacheArgs ca{};
  ca.len = s->bs_len.as<int>(); ca.parent = s->bs_parent.as<int>(); ca.parent_draft = s->bs_parent_draft.as<int>();
  ca.prev_len = s->t_prev_len.as<int>(); ca.active = s->bs_active.as<uint8_t>();
  // ONE cache buffer and a candidate -> slot map (k_bsp_select): most children append in their parent's slot, few copy
  ca.k_old = s->tk[0].as<float>(); ca.v_old = s->tv[0].as<float>(); ca.k_new = s->tk[0].as<float>(); ca.v_new = s->tv[0].as<float>();
  ca.slot_parent = j.a.cache_slot_parent; ca.slot_self = j.a.cache_slot;
  ca.cache_layer_stride = cache_layer; ca.cache_seq_stride = cache_seq;
  ca.qkv_prev = s->qkv.as<float>(); ca.qkv_layer_stride = (long long)MC * step_rps(j.N, dl) * 3 * d;
  ca.prev_slot_of = s->t_slot_of.as<int>(); ca.prev_N = j.N; ca.prev_D = dl; ca.d = d;
  hipLaunchKernelGGL(k_tree_cache, dim3(MC, Ld), dim3(256), 0, st, ca);      // fresh candidates (parent -1) have nothing to inherit
  HIP_TRY(hipGetLastError());
  BeamListArgs la{};
  la.active = s->bs_active.as<uint8_t>(); la.per_cand = s->bs_per_cand.as<int>(); la.len = s->bs_len.as<int>();
  la.n_cand = MC; la.N = j.N; la.dl = dl;
  la.act_idx = s->act_idx.as<int>(); la.slot_of = s->t_slot_of.as<int>(); la.prev_len = s->t_prev_len.as<int>();
  la.st = s->state.as<DecState>(); la.cnt = s->bs_cnt.as<BeamCounters>(); la.summary = s->beam_summary.as<int>();
  hipLaunchKernelGGL(k_bs_list, dim3(1), dim3(256), 0, st, la);
  HIP_TRY(hipGetLastError());
  StepCtx k{};
  k.B = MC; k.Ls = j.Ls_cap; k.N = j.N; k.D = dl; k.Lc = j.Lc; k.gen_ld = j.gen_ld; k.max_len = j.p.max_len;
  k.kcache = s->tk[0].as<float>(); k.vcache = s->tv[0].as<float>(); k.cache_slot = j.a.cache_slot;
  k.src_of = s->t_src_of.as<int>(); k.src_len = s->bp_cand_len.as<int>();
  k.want_argmax = false; k.variant = variant;
  TTX_TRY(run_step(s, st, k, std::min(j.p.max_len, ((j.p.max_len + 63) / 64) * 64)));
  BeamHitsArgs ha{};
  ha.logits = s->logits.as<float>(); ha.V = V; ha.finished = s->bs_fin.as<uint8_t>(); ha.slot_of = s->t_slot_of.as<int>();
  ha.per_cand = s->bs_per_cand.as<int>(); ha.drafts32 = s->drafts.as<int>();
  ha.n_cand = MC; ha.N = j.N; ha.dl = dl; ha.K = j.K; ha.nucleus = 0.9975f; ha.hit = s->bs_hit.as<uint8_t>(); ha.dl_of = j.a.cand_dl;
  BeamLeaves2Args le{};
  le.logits = s->logits.as<float>(); le.V = V; le.finished = s->bs_fin.as<uint8_t>(); le.slot_of = s->t_slot_of.as<int>();
  le.per_cand = s->bs_per_cand.as<int>(); le.drafts32 = s->drafts.as<int>(); le.logp = s->bs_logp.as<float>();
  le.hit = s->bs_hit.as<uint8_t>(); le.cnt = s->bs_cnt.as<BeamCounters>();
  le.n_cand = MC; le.N = j.N; le.dl = dl; le.K = j.K; le.bos = j.p.bos_token; le.pad = j.p.pad_token; le.smart = j.smart ? 1 : 0;
  le.best_n = s->bs_best_n.as<int>(); le.best_slot = s->bs_best_slot.as<int>(); le.chosen = s->bs_chosen.as<int64_t>();
  le.leaf_score = s->leaf_score.as<float>(); le.leaf_tok = s->leaf_tok.as<int>(); le.leaf_cnt = s->leaf_cnt.as<int>();
  le.live = s->bs_mark.as<uint8_t>(); le.dl_of = j.a.cand_dl; le.grp_of = j.a.bat_grp; le.cand_batch = j.a.cand_batch;
  const dim3 hits_grid(MC, cdiv(std::max(j.N * dl, 1), BS_HITS_WAVES));
  const size_t leaves_lds = (size_t)2 * (dl + 1) * 4;
  if (V <= 256) {
    hipLaunchKernelGGL(k_bs_hits<4>, hits_grid, dim3(BS_HITS_WAVES * 64), 0, st, ha);
    hipLaunchKernelGGL(k_bs_leaves<4>, dim3(MC), dim3(BS_LEAVES_THREADS), leaves_lds, st, le);
  } else if (V <= 512) {
    hipLaunchKernelGGL(k_bs_hits<8>, hits_grid, dim3(BS_HITS_WAVES * 64), 0, st, ha);
    hipLaunchKernelGGL(k_bs_leaves<8>, dim3(MC), dim3(BS_LEAVES_THREADS), leaves_lds, st, le);
  } else {
    hipLaunchKernelGGL(k_bs_hits<NUC_VPL>, hits_grid, dim3(BS_HITS_WAVES * 64), 0, st, ha);
    hipLaunchKernelGGL(k_bs_leaves<NUC_VPL>, dim3(MC), dim3(BS_LEAVES_THREADS), leaves_lds, st, le);
  }
  HIP_TRY(hipGetLastError());
  const size_t lds = 2 * (size_t)j.K * (dl + 1) * j.K * 4;
  if (lds > 64 * 1024 && !s->attr_pool_select) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bsp_select), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    s->attr_pool_select = true;
  }
  hipLaunchKernelGGL(k_bsp_select, dim3(j.C), dim3(256), lds, st, j.a);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_bsp_batches, dim3(cdiv(j.C, 256)), dim3(256), 0, st, j.a);
  hipLaunchKernelGGL(k_bsp_retire, dim3(j.C), dim3(256), 0, st, j.a);
  hipLaunchKernelGGL(k_bsp_publish, dim3(cdiv(j.C, 256)), dim3(256), 0, st, j.a);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

static int bpool_launch_iter(BeamPoolJob& j, int n_running) {
  ttx_session* s = j.s;
  const int cur = j.cur;
  const int variant = variant_for_rows(s, (long long)std::max(1, n_running) * step_rps(j.N, j.D0), true);
  int rc = TTX_OK;
  if (!s->use_graphs || s->profile) {
    rc = bpool_enqueue_iter(j, cur, variant);
  } else {
    s->graphs_current();
    const std::vector<int> key{-7, j.C, j.Ls_cap, j.K, j.N, j.D0, j.smart ? 1 : 0, j.p.max_len, cur, variant, j.p.pad_token, j.p.bos_token,
                               j.p.eos_token, j.p.replace_token, j.p.max_steps};
    auto it = s->beam_graphs.find(key);
    if (it == s->beam_graphs.end() && !s->beam_warmed.count(key)) {
      s->beam_warmed.insert(key);                  // first use runs eagerly: function attributes are set outside capture
      rc = bpool_enqueue_iter(j, cur, variant);
    } else {
      if (it == s->beam_graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_TRY(hipStreamBeginCapture(j.st, hipStreamCaptureModeThreadLocal));
        rc = bpool_enqueue_iter(j, cur, variant);
        hipError_t e = hipStreamEndCapture(j.st, &graph);
        if (rc != TTX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return fail(TTX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        if (s->beam_graphs.size() > 256) s->drop_graphs();
        it = s->beam_graphs.emplace(key, exec).first;
      }
      HIP_TRY(hipGraphLaunch(it->second, j.st));
    }
  }
  TTX_TRY(rc);
  ++j.launched;
  j.cur = cur ^ 1;
  j.admitted_since = 0;
  j.last_progress = std::chrono::steady_clock::now();
  j.idle_spins = 0;
  return TTX_OK;
}

extern "C" int ttx_beam_speculative_generate_pool(ttx_session** sessions, int n_sessions, const int64_t* d_src, int R_total, int Ls_all,
                                                  const int32_t* h_len, const int32_t* h_batch_of, int n_batches,
                                                  const int32_t* h_given_ls, int capacity, const ttx_beam_params* p, int64_t* d_out,
                                                  int16_t* d_trace_len, int32_t* d_summary, int trace_cap, ttx_beam_stats* stats,
                                                  void* stream) {
  if (!sessions || n_sessions <= 0 || !d_src || R_total < 0 || !h_len || !h_batch_of || n_batches < 0 || !h_given_ls || capacity <= 0 ||
      capacity > 1024 || !p || !d_out || !d_trace_len || !d_summary || trace_cap <= 0 || trace_cap > 32000 || !stats)
    return fail(TTX_ERR_INVALID, "bad argument to ttx_beam_speculative_generate_pool");
  if (R_total == 0) return TTX_OK;
  for (int i = 0; i < n_sessions; ++i) { if (!sessions[i]) return fail(TTX_ERR_INVALID, "null session"); TTX_TRY(session_alive(sessions[i])); }
  // the work list: whole batches, in order; batch b owns the sources [first[b], first[b + 1])
  std::vector<int> first(n_batches + 1, 0);
  int len_max = 2, max_batch = 0;
  for (int i = 0, b = -1; i < R_total; ++i) {
    if (h_batch_of[i] != b) {
      if (h_batch_of[i] != b + 1 || h_batch_of[i] >= n_batches) return fail(TTX_ERR_INVALID, "h_batch_of must number the batches 0, 1, ... in order");
      b = h_batch_of[i];
      first[b] = i;
    }
    first[b + 1] = i + 1;
    if (h_len[i] < 2 || h_len[i] > Ls_all) return fail(TTX_ERR_INVALID, "row length outside [2, width of the source matrix]");
    if (h_given_ls[b] < h_len[i]) return fail(TTX_ERR_INVALID, "a batch's given width is smaller than one of its sources");
    if (p->smart_drafts_mode && h_given_ls[b] - 5 <= 0) return fail(TTX_ERR_REFERENCE, "The number of drafts must be greater than 0");
    len_max = std::max(len_max, (int)h_len[i]);
  }
  if (R_total > 0 && h_batch_of[R_total - 1] != n_batches - 1) return fail(TTX_ERR_INVALID, "n_batches does not match h_batch_of");
  for (int b = 0; b < n_batches; ++b) max_batch = std::max(max_batch, first[b + 1] - first[b]);
  if (max_batch > capacity) return fail(TTX_ERR_INVALID, "a batch has more sources than the pool has slots");
  const ttx_config& c = sessions[0]->m->cfg;
  int Ls_cap = std::min(std::max(192, ((len_max + 63) / 64) * 64), c.max_positions);
  Ls_cap = std::max(Ls_cap, len_max);
  TTX_TRY(beam_validate(sessions[0], d_src, 1, Ls_cap, p, d_out));
  HIP_TRY(hipSetDevice(sessions[0]->m->device));
  release_retired();
  EventGuard ready;
  HIP_TRY(hipEventCreateWithFlags(&ready.e, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ready.e, (hipStream_t)stream));
  const int n_jobs = std::max(1, std::min({n_sessions, n_batches, cdiv(R_total, std::max(1, std::min(capacity / 2, 8)))}));
  const int C = std::min(capacity, std::max(max_batch, cdiv(R_total, n_jobs)));
  BeamPoolIo io{};
  io.out = d_out; io.trace_len = d_trace_len; io.summary = d_summary; io.T_cap = trace_cap;
  std::vector<BeamPoolJob> jobs(n_jobs);
  int rc_final = TTX_OK;
  for (int i = 0; i < n_jobs && rc_final == TTX_OK; ++i) {
    ttx_session* s = sessions[i];
    TTX_TRY(ensure_own_stream(s));
    HIP_TRY(hipStreamWaitEvent(s->own_stream, ready.e, 0));
    rc_final = bpool_start(jobs[i], s, s->own_stream, C, Ls_cap, p, io, h_len, h_batch_of, h_given_ls, R_total, n_batches);
  }
  const int min_admit = std::max(1, C / 4);                    // admissions of at least a quarter pool (or into an empty one)
  const bool serial = sessions[0]->profile;
  int cursor_b = 0, done = 0;                                  // next batch of the work list
  const auto t_call = std::chrono::steady_clock::now();
  bool hung = false;
  ttx_beam_stats acc{};
  while (done < n_jobs && rc_final == TTX_OK) {
    bool progressed = false;
    int only = -1;                       // profiling sessions: one pool after another (see the greedy pool)
    if (serial)
      for (int i = n_jobs - 1; i >= 0; --i)
        if (jobs[i].phase != 3) only = i;
    for (int i = 0; i < n_jobs && rc_final == TTX_OK; ++i) {
      if (serial && i != only) continue;
      BeamPoolJob& j = jobs[i];
      ttx_session* s = j.s;
      volatile BeamPoolHost* bh = s->bp_host;
      if (j.phase == 1) {
        if (j.launched > 0 && bh->steps_done < j.launched) {                  // the iteration in flight has not published yet
          if ((++j.idle_spins & 0xffff) == 0 && watchdog_expired(j.last_progress)) { rc_final = session_hung(s); hung = true; break; }
          continue;
        }
        if (bh->error) { rc_final = fail(TTX_ERR_HIP, "batch pool bookkeeping failed (admitted more sources than free slots)"); break; }
        int n_live = (j.launched == 0 ? 0 : bh->n_live) + j.admitted_since;
        int n_running = (j.launched == 0 ? 0 : bh->n_running) + j.admitted_since;
        const int free_slots = C - n_live;
        if (serial && j.quota_b < 0) {                  // serial pass: an equal contiguous share of the rest of the list (greedy pool)
          const int target = first[cursor_b] + cdiv(R_total - first[cursor_b], n_jobs - i);
          j.quota_b = cursor_b;
          while (j.quota_b < n_batches && first[j.quota_b] < target) ++j.quota_b;
        }
        const int list_end_b = serial ? j.quota_b : n_batches;
        if (cursor_b < list_end_b && (n_live == 0 || free_slots >= std::max(min_admit, first[cursor_b + 1] - first[cursor_b]))) {
          // whole batches, as many as fit; the last batches of the list are shared out over the pools still running, so that
          // they drain together, and a first fill that fits the pools at once is split evenly
          int n_run_jobs = 0;
          for (const BeamPoolJob& o : jobs) n_run_jobs += (o.phase == 1);
          const int remaining = first[list_end_b] - first[cursor_b];
          int share = std::max(1, cdiv(remaining, std::max(1, serial ? 1 : n_run_jobs)));
          if (!serial && j.launched == 0 && (long long)n_jobs * C >= R_total) share = std::max(1, cdiv(remaining, n_jobs - i));
          int take = 0, b_end = cursor_b;
          while (b_end < list_end_b) {
            const int sz = first[b_end + 1] - first[b_end];
            if (take + sz > free_slots || (take > 0 && take + sz > share)) break;
            take += sz;
            ++b_end;
          }
          if (take > 0) {
            const int r_first = first[cursor_b];
            int Ls_new = 2;
            for (int r = r_first; r < r_first + take; ++r) Ls_new = std::max(Ls_new, (int)h_len[r]);
            rc_final = bpool_admit(j, d_src + (size_t)r_first * Ls_all, Ls_all, take, Ls_new, r_first);
            if (rc_final != TTX_OK) break;
            cursor_b = b_end;
            n_live += take;
            n_running += take;
          }
        }
        if (n_live == 0) {
          if (hipEventRecord(s->ev_c, j.st) != hipSuccess ||
              hipMemcpyAsync(s->host_state, s->bs_cnt.p, sizeof(BeamCounters), hipMemcpyDeviceToHost, j.st) != hipSuccess ||
              hipEventRecord(s->ev_done, j.st) != hipSuccess) {
            rc_final = fail(TTX_ERR_HIP, "batch pool: enqueueing the final counter read-back failed");
            break;
          }
          j.phase = 2;
        } else {
          if (j.launched > (long long)(trace_cap + 2) * (R_total + 1)) { rc_final = fail(TTX_ERR_HIP, "batch pool failed to terminate"); break; }
          if (s->host_timing)
            fprintf(stderr, "[ttx pool %d] t=%.3f ms iteration %d: %d live sources, %d running candidates, next batch %d of %d\n", i,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), j.launched + 1, n_live,
                    n_running, cursor_b, n_batches);
          rc_final = bpool_launch_iter(j, n_running);
        }
        progressed = true;
      } else if (j.phase == 2) {
        if (hipEventQuery(s->ev_done) == hipSuccess) {
          const BeamCounters* cn = reinterpret_cast<const BeamCounters*>(s->host_state);
          acc.model_calls += cn->model_calls;
          acc.input_lines += cn->input_lines;
          acc.running_rows += cn->running_rows;
          acc.verified_positions += cn->verified_positions;
          acc.executed_positions += cn->executed_positions;
          acc.kv_prefix_positions += cn->kv_prefix_positions;
          acc.running_candidates += cn->running_cands;
          acc.src_tokens_padded += j.src_tokens_padded;
          float ms = 0.f;
          if (hipEventElapsedTime(&ms, s->ev_a, s->ev_c) == hipSuccess) acc.decode_ms += ms;
          collect_gemm_profile(s, j.st);
          j.phase = 3;
          ++done;
          progressed = true;
        }
      }
    }
    if (!progressed) __builtin_ia32_pause();
  }
  if (hung) {
    for (int i = 0; i < n_jobs; ++i) jobs[i].s->dead = true;
  } else {
    for (int i = 0; i < n_jobs; ++i)
      if (jobs[i].s && jobs[i].s->own_stream) (void)hipStreamSynchronize(jobs[i].s->own_stream);
  }
  acc.status = rc_final;
  *stats = acc;
  return rc_final;
}

// ------------------------------------------------------------------------------------------------
// Standard beam search, whole loop native (standard_decoding.py:89-174): per-hypothesis KV cache (the tree kernels),
// the verify-step kernels with one row per running hypothesis, k_beam_step for log-softmax + top-beam + row assembly.
static int beam_generate_impl(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_search_params* p, int64_t* d_out,
                              ttx_beam_search_stats* stats, void* stream);

extern "C" int ttx_beam_generate(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_search_params* p, int64_t* d_out,
                                 ttx_beam_search_stats* stats, void* stream) {
  const int rc = beam_generate_impl(s, d_src, B, Ls, p, d_out, stats, stream);
  // an error return must not leave work of this call in flight on the session's stream (the caller may free d_src / d_out)
  if (rc != TTX_OK && s && !s->dead && s->own_stream) (void)hipStreamSynchronize(s->own_stream);
  return rc;
}

static int beam_generate_impl(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_search_params* p, int64_t* d_out,
                              ttx_beam_search_stats* stats, void* stream) {
  if (!s || !d_src || !p || !d_out || !stats || B <= 0 || Ls <= 1) return fail(TTX_ERR_INVALID, "bad argument to ttx_beam_generate");
  TTX_TRY(session_alive(s));
  const ttx_model* m = s->m;
  const ttx_config& c = m->cfg;
  const int K = p->beam_size, max_len = p->max_len, V = c.vocab_size;
  if (max_len <= 1) return fail(TTX_ERR_REFERENCE, "assert self.max_len > 1 (standard_decoding.py:79)");
  if (K <= 0) return fail(TTX_ERR_REFERENCE, "assert self.beam_size > 0 (standard_decoding.py:80)");
  if (K > V) return fail(TTX_ERR_REFERENCE, "beam_size larger than the vocabulary: topk(k) of the first step raises");
  if ((size_t)K * V * 4 > 150 * 1024) return fail(TTX_ERR_INVALID, "beam_size x vocabulary beyond the selection kernel's LDS image");
  if (p->pad_token != c.pad_token) return fail(TTX_ERR_INVALID, "generator pad token differs from the model's");
  if (max_len + 2 > c.max_positions) return fail(TTX_ERR_INVALID, "max_len exceeds the positional table");
  HIP_TRY(hipSetDevice(m->device));
  release_retired();
  TTX_TRY(ensure_own_stream(s));
  s->ev_used = 0;                     // profiling sessions: the event pairs of this call start from the first one
  HIP_TRY(hipEventRecord(s->ev_done, (hipStream_t)stream));
  HIP_TRY(hipStreamWaitEvent(s->own_stream, s->ev_done, 0));
  hipStream_t st = s->own_stream;
  const int d = c.embedding_dim, Ld = c.num_decoder_layers;
  const int MC = B * K, ld = max_len + 2, Lc = max_len + 2;
  const size_t Mmax = (size_t)MC;
  int rc = TTX_OK;
  auto need = [&](Buf& b, size_t bytes) { if (rc == TTX_OK) rc = ensure(b, bytes, st); };
  need(s->tok_src, (size_t)B * Ls * 4); need(s->src_valid, (size_t)B * Ls); need(s->memory, (size_t)B * Ls * d * 4);
  need(s->memkv, (size_t)B * Ls * Ld * 2 * d * 4);
  need(s->drafts, (size_t)MC * 4);
  need(s->gen, (size_t)MC * ld * 4); need(s->front, (size_t)MC * 4); need(s->act_idx, (size_t)MC * 4);
  need(s->pred, Mmax * 4); need(s->state, sizeof(DecState)); need(s->logits, Mmax * V * 4);
  for (int i = 0; i < 2; ++i) { need(s->tk[i], (size_t)Ld * MC * Lc * d * 4); need(s->tv[i], (size_t)Ld * MC * Lc * d * 4); }
  need(s->t_prev_len, (size_t)MC * 4); need(s->t_slot_of, (size_t)MC * 4); need(s->t_src_of, (size_t)MC * 4);
  need(s->bs_cand_next, (size_t)MC * ld * 8); need(s->bs_len_next, (size_t)MC * 4); need(s->bs_fin_next, (size_t)MC);
  need(s->bs_logp_next, (size_t)MC * 4); need(s->bs_len, (size_t)MC * 4); need(s->bs_fin, (size_t)MC); need(s->bs_active, (size_t)MC);
  need(s->bs_logp, (size_t)MC * 4); need(s->bs_per_cand, (size_t)MC * 4); need(s->bs_parent, (size_t)MC * 4);
  need(s->bs_parent_draft, (size_t)MC * 4); need(s->bs_cnt, sizeof(BeamCounters)); need(s->beam_summary, 8 * 4);
  const size_t Macts = std::max(Mmax, (size_t)B * Ls);
  if (rc == TTX_OK) rc = ensure_acts(s, st, Macts, 1);
  need(s->qkv, std::max((size_t)Ld * Mmax, (size_t)B * Ls) * 3 * d * 4);
  need(s->slab, sizeof(float) * 16 * Macts * d);
  s->graphs_current();
  TTX_TRY(rc);
  if (!s->beam_host && hipHostMalloc((void**)&s->beam_host, sizeof(BeamHost), hipHostMallocMapped) != hipSuccess)
    return fail(TTX_ERR_NOMEM, "hipHostMalloc failed");
  std::memset(s->beam_host, 0, sizeof(BeamHost));
  BeamHost* dev_host = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dev_host, (void*)s->beam_host, 0));
  // self.model(src, y) of the first step = encoder + the decoder on <BOS>; the reference then re-encodes the source once per
  // beam (:120-124): identical rows, so the beams of a source share its memory row here
  TTX_TRY(prepare_tokens(st, d_src, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B * Ls, c.pad_token));
  TTX_TRY(run_encoder(s, st, s->tok_src.as<int>(), s->src_valid.as<uint8_t>(), B, Ls, s->memory.as<float>()));
  const int gv = variant_for_rows(s, (long long)B * Ls, false);
  TTX_TRY(launch_gemm(s, st, s->memory.as<float>(), d, m->p(m->cross_kv_w), d, m->p(m->cross_kv_b), s->memkv.as<float>(),
                      Ld * 2 * d, nullptr, B * Ls, Ld * 2 * d, d, false, 0, 0, gv));
  hipLaunchKernelGGL(k_bs_init, dim3(64), dim3(256), 0, st, s->bs_cand_next.as<int64_t>(), ld, s->bs_len_next.as<int>(),
                     s->bs_fin_next.as<uint8_t>(), s->bs_logp_next.as<float>(), s->bs_parent.as<int>(), s->bs_parent_draft.as<int>(),
                     MC, B, p->bos_token, p->pad_token, s->bs_cnt.as<BeamCounters>());
  HIP_TRY(hipGetLastError());

  const long long cache_seq = (long long)Lc * d, cache_layer = (long long)MC * cache_seq;
  int n_cand = B, beam = 1, width = 1, cur = 0, launched = 0, n_eos = 0;
  const int max_iters = max_len - 1;                 // the <BOS> step plus `predictions - 1` loop iterations (:127-131)
  auto enqueue = [&](bool first) -> int {
    BeamPrepArgs pa{};
    pa.cand_next = s->bs_cand_next.as<int64_t>(); pa.ld = ld; pa.len_next = s->bs_len_next.as<int>();
    pa.fin_next = s->bs_fin_next.as<uint8_t>(); pa.logp_next = s->bs_logp_next.as<float>();
    pa.n_cand = n_cand; pa.beam = beam; pa.dl = 0; pa.N = 1; pa.pad = p->pad_token; pa.smart = 0;
    pa.gen = s->gen.as<int>(); pa.front = s->front.as<int>(); pa.len = s->bs_len.as<int>(); pa.active = s->bs_active.as<uint8_t>();
    pa.finished = s->bs_fin.as<uint8_t>(); pa.logp = s->bs_logp.as<float>(); pa.per_cand = s->bs_per_cand.as<int>();
    pa.drafts32 = s->drafts.as<int>();
    hipLaunchKernelGGL(k_bs_prep, dim3(MC), dim3(256), 0, st, pa);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_bs_src_of, dim3(cdiv(MC, 256)), dim3(256), 0, st, s->t_src_of.as<int>(), MC, beam);
    HIP_TRY(hipGetLastError());
    const int nxt = cur ^ 1;
    if (!first) {
      TreeCacheArgs ca{};
      ca.len = s->bs_len.as<int>(); ca.parent = s->bs_parent.as<int>(); ca.parent_draft = s->bs_parent_draft.as<int>();
      ca.prev_len = s->t_prev_len.as<int>(); ca.active = s->bs_active.as<uint8_t>();
      ca.k_old = s->tk[cur].as<float>(); ca.v_old = s->tv[cur].as<float>(); ca.k_new = s->tk[nxt].as<float>(); ca.v_new = s->tv[nxt].as<float>();
      ca.cache_layer_stride = cache_layer; ca.cache_seq_stride = cache_seq;
      ca.qkv_prev = s->qkv.as<float>(); ca.qkv_layer_stride = (long long)MC * 3 * d;
      ca.prev_slot_of = s->t_slot_of.as<int>(); ca.prev_N = 1; ca.prev_D = 0; ca.d = d;
      hipLaunchKernelGGL(k_tree_cache, dim3(MC, Ld), dim3(256), 0, st, ca);
      HIP_TRY(hipGetLastError());
    }
    BeamListArgs la{};
    la.active = s->bs_active.as<uint8_t>(); la.per_cand = s->bs_per_cand.as<int>(); la.len = s->bs_len.as<int>();
    la.n_cand = n_cand; la.N = 1; la.dl = 0;
    la.act_idx = s->act_idx.as<int>(); la.slot_of = s->t_slot_of.as<int>(); la.prev_len = s->t_prev_len.as<int>();
    la.st = s->state.as<DecState>(); la.cnt = s->bs_cnt.as<BeamCounters>(); la.summary = s->beam_summary.as<int>();
    hipLaunchKernelGGL(k_bs_list, dim3(1), dim3(256), 0, st, la);
    HIP_TRY(hipGetLastError());
    StepCtx k{};
    k.B = MC; k.Ls = Ls; k.N = 1; k.D = 0; k.Lc = Lc; k.gen_ld = ld; k.max_len = max_len;
    k.kcache = s->tk[nxt].as<float>(); k.vcache = s->tv[nxt].as<float>(); k.src_of = s->t_src_of.as<int>(); k.want_argmax = false;
    k.variant = variant_for_rows(s, (long long)std::max(1, n_cand - n_eos), true);
    TTX_TRY(run_step(s, st, k, std::min(max_len, ((width + 63) / 64) * 64)));
    BeamStepArgs sa{};
    sa.logits = s->logits.as<float>(); sa.V = V; sa.slot_of = s->t_slot_of.as<int>(); sa.finished = s->bs_fin.as<uint8_t>();
    sa.score = s->bs_logp.as<float>(); sa.gen = s->gen.as<int>(); sa.ld = ld; sa.width = width;
    sa.B = B; sa.beam = beam; sa.K = K; sa.pad = p->pad_token; sa.eos = p->eos_token;
    sa.new_cand = s->bs_cand_next.as<int64_t>(); sa.new_score = s->bs_logp_next.as<float>(); sa.parent = s->bs_parent.as<int>();
    sa.new_len = s->bs_len_next.as<int>(); sa.new_finished = s->bs_fin_next.as<uint8_t>(); sa.parent_draft = s->bs_parent_draft.as<int>();
    sa.summary = s->beam_summary.as<int>();
    const size_t lds = (size_t)beam * V * 4;
    if (lds > 64 * 1024 && !s->attr_step) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_beam_step), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      s->attr_step = true;
    }
    hipLaunchKernelGGL(k_beam_step, dim3(B), dim3(256), lds, st, sa);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_bs_publish, dim3(1), dim3(64), 0, st, s->beam_summary.as<int>(), dev_host, s->bs_cnt.as<BeamCounters>());
    HIP_TRY(hipGetLastError());
    return TTX_OK;
  };
  volatile BeamHost* bh = s->beam_host;
  while (launched < max_iters) {
    const bool first = launched == 0;
    // width changes every step, so a step's graph would be replayed once: the steps run eagerly (about 12 launches per
    // decoder layer, enqueued well ahead of the device)
    TTX_TRY(enqueue(first));
    ++launched;
    cur ^= 1;
    unsigned spins = 0;
    const auto since = std::chrono::steady_clock::now();
    while (bh->steps_done < launched) {
      if ((++spins & 0xffff) == 0 && watchdog_expired(since)) return session_hung(s);
      __builtin_ia32_pause();
    }
    width += 1;
    n_cand = B * K; beam = K;
    n_eos = bh->summary[0];
    if (n_eos == B * K && !first) break;                   // :166 (the check sits inside the loop, after the first step)
  }
  HIP_TRY(hipMemcpy2DAsync(d_out, (size_t)max_len * 8, s->bs_cand_next.p, (size_t)ld * 8, (size_t)width * 8, (size_t)B * K,
                           hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->host_state, s->bs_cnt.p, sizeof(BeamCounters), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const BeamCounters* cn = reinterpret_cast<const BeamCounters*>(s->host_state);
  stats->model_calls = cn->model_calls;
  stats->running_rows = cn->running_cands;
  stats->out_width = width;
  return TTX_OK;
}

// Parity instrumentation: the verify step selected by ttx_gen_params.want_logits (1-based step number) of the most
// recent generate call on this session: its pre-argmax logits and the loop state they were computed from.
// All destinations are HOST pointers; sizes: logits [n_active*rps, V], act [n_active], front [B], gen [B, gen_ld].
// info[0..5] = n_active, rps (rows per sequence = 1 + N*D), B, gen_ld, V, step.  Pass nulls to query info only.
extern "C" int ttx_debug_step_snapshot(ttx_session* s, int32_t* info, float* h_logits, int32_t* h_act, int32_t* h_front,
                                       int32_t* h_gen) {
  if (!s || !info) return fail(TTX_ERR_INVALID, "null argument to ttx_debug_step_snapshot");
  if (s->snap_step == 0) return fail(TTX_ERR_INVALID, "no verify step was recorded (set ttx_gen_params.want_logits)");
  HIP_TRY(hipSetDevice(s->m->device));
  HIP_TRY(hipDeviceSynchronize());
  DecState st;
  HIP_TRY(hipMemcpy(&st, s->snap_state.p, sizeof(DecState), hipMemcpyDeviceToHost));
  const int V = s->m->cfg.vocab_size;
  info[0] = st.n_active; info[1] = s->snap_rps; info[2] = s->snap_B; info[3] = s->snap_gen_ld; info[4] = V; info[5] = s->snap_step;
  if (h_logits) HIP_TRY(hipMemcpy(h_logits, s->snap_logits.p, (size_t)st.n_active * s->snap_rps * V * 4, hipMemcpyDeviceToHost));
  if (h_act) HIP_TRY(hipMemcpy(h_act, s->snap_act.p, (size_t)st.n_active * 4, hipMemcpyDeviceToHost));
  if (h_front) HIP_TRY(hipMemcpy(h_front, s->snap_front.p, (size_t)s->snap_B * 4, hipMemcpyDeviceToHost));
  if (h_gen) HIP_TRY(hipMemcpy(h_gen, s->snap_gen.p, (size_t)s->snap_B * s->snap_gen_ld * 4, hipMemcpyDeviceToHost));
  return TTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-side tokenizer / collate / detokenizer (no GPU)
extern "C" int ttx_tokenizer_create(const char* const* tokens, const int32_t* ids, int n, ttx_tokenizer** out) {
  if (!tokens || !ids || n <= 0 || !out) return fail(TTX_ERR_INVALID, "bad argument to ttx_tokenizer_create");
  ttx_tokenizer* t = new ttx_tokenizer();
  int32_t max_id = -1;
  for (int i = 0; i < n; ++i) max_id = std::max(max_id, ids[i]);
  t->dec.assign((size_t)max_id + 1, std::string());
  for (int i = 0; i < n; ++i) {
    if (!tokens[i] || ids[i] < 0) { delete t; return fail(TTX_ERR_INVALID, "vocabulary entry with a null token or negative id"); }
    t->enc[tokens[i]] = ids[i];
    t->dec[ids[i]] = tokens[i];
  }
  *out = t;
  return TTX_OK;
}

extern "C" void ttx_tokenizer_destroy(ttx_tokenizer* t) { delete t; }

// ChemSMILESTokenizer.encode: returns the number of ids the line needs (BOS and EOS included); writes min(that, cap).
extern "C" int ttx_tokenizer_encode(const ttx_tokenizer* t, const char* line, int32_t* out, int cap) {
  if (!t || !line) return fail(TTX_ERR_INVALID, "null argument to ttx_tokenizer_encode");
  int n = 0;
  auto put = [&](int32_t id) { if (out && n < cap) out[n] = id; ++n; };
  put(t->bos);
  std::string piece;
  ttxtok::split(line, std::strlen(line), [&](const char* p, size_t len) {
    piece.assign(p, len);
    auto it = t->enc.find(piece);
    put(it == t->enc.end() ? t->unk : it->second);
  });
  put(t->eos);
  return n;
}

// Tokenize B lines and pad them to the longest (the DataModule's collate: seq2seq_wrappers.py:121-127).  `out` is int64
// [B, cap_cols] row-major (HOST); returns the padded width (<= cap_cols) or, if some line needs more columns, minus the
// width required (nothing usable is written then).
extern "C" int ttx_tokenizer_encode_batch(const ttx_tokenizer* t, const char* const* lines, int B, int64_t* out, int cap_cols) {
  if (!t || !lines || B <= 0 || !out || cap_cols <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_tokenizer_encode_batch");
  std::vector<int32_t> row((size_t)cap_cols);
  int width = 0;
  for (int b = 0; b < B; ++b) {
    const int n = ttx_tokenizer_encode(t, lines[b], row.data(), cap_cols);
    if (n < 0) return n;
    width = std::max(width, n);
    if (n <= cap_cols) {
      for (int i = 0; i < n; ++i) out[(size_t)b * cap_cols + i] = row[i];
      for (int i = n; i < cap_cols; ++i) out[(size_t)b * cap_cols + i] = t->pad;
    }
  }
  return width <= cap_cols ? width : -width;
}

// GenericTokenizer.decode with skip_service_tokens=True: returns the string length (NUL excluded) the ids need; writes at
// most cap-1 characters and a terminating NUL.  Ids outside the vocabulary fail (the reference raises KeyError).
extern "C" int ttx_tokenizer_decode(const ttx_tokenizer* t, const int64_t* ids, int n, char* out, int cap) {
  if (!t || (!ids && n > 0)) return fail(TTX_ERR_INVALID, "null argument to ttx_tokenizer_decode");
  size_t len = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t id = ids[i];
    if (id != t->bos && id != t->eos && id != t->pad) {
      if (id < 0 || (size_t)id >= t->dec.size() || (t->dec[id].empty() && t->enc.find("") == t->enc.end()))
        return fail(TTX_ERR_REFERENCE, "token id outside the vocabulary (KeyError in the reference)");
      const std::string& tok = t->dec[id];
      for (char ch : tok) { if (out && (int)len + 1 < cap) out[len] = ch; ++len; }
    }
    if (id == t->eos) break;
  }
  if (out && cap > 0) out[std::min(len, (size_t)cap - 1)] = '\0';
  return (int)len;
}

// Development aid (tools/bench_gemm.py): one GEMM shape in isolation (ttx_gemm.hip: gemm_bench).
extern "C" int ttx_debug_gemm_bench(ttx_session* s, int M, int N, int K, int splits, int variant, int reps, double* us_per_launch,
                                    double* max_abs_diff) {
  return gemm_bench(s, M, N, K, splits, variant, reps, us_per_launch, max_abs_diff);
}

extern "C" int ttx_last_kernel_profile(ttx_session* s, double* gemm_ms, int64_t* gemm_launches, double* empty_pair_ms) {
  if (!s) return fail(TTX_ERR_INVALID, "null session");
  if (gemm_ms) *gemm_ms = s->prof_ms;
  if (gemm_launches) *gemm_launches = s->prof_launches;
  s->prof_ms = 0;
  s->prof_launches = 0;
  if (empty_pair_ms) *empty_pair_ms = s->prof_empty_pair_ms < 0 ? 0.0 : s->prof_empty_pair_ms;
  return TTX_OK;
}
