// Whole-model plan: the native runtime under AMCTransformer.forward / loss.backward().
//
// Host-side C++ that lays out parameters, shadows and the activation workspace and issues the
// kernel sequence of one forward / backward on a HIP stream.  No allocation, no synchronisation:
// the caller (PyTorch-ROCm host code) owns every buffer and may capture the calls in a hipGraph.
//
// Reference call stack this replaces (under /root/reference/Transformer_Thesis/):
//   AMCTransformer.forward   ViT/models/amc_transformer.py:26-31, transformer_rawIQ/models/transformer_rawIQ.py:72-98
//   Encoder.forward          ViT/models/encoder.py:34-53, transformer_rawIQ/models/encoder.py:86-117
//   EncoderLayer.forward     ViT/models/blocks/encoder_layer.py:18-35   (post-norm)
// and the autograd backward of all of it.
//
// Mixed precision policy (the reference is fp32; this is the build's stated policy):
//   fp32 master parameters / gradients / LN statistics / softmax / logits / loss,
//   bf16 activations, activation gradients and GEMM operands (bf16 weight shadows, refreshed after
//   every optimizer step), fp32 MFMA accumulation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

struct Entry {
  std::string name;
  size_t off;
  int ndim;
  int dims[4];
};

struct LayerOff {
  size_t wqkv, bqkv, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2, end;
  // transposed bf16 shadows (byte offsets into the shadow buffer)
  size_t t_wqkv, t_wo, t_w1, t_w2;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsPlan {
  size_t step_ctr, patches, x0, head_feat, head_stat;
  struct L { size_t qkv, att, lse, z1, mean1, rstd1, x1, hid, gate, z2, mean2, rstd2, x2; };
  std::vector<L> layers;
  size_t gA, gB, gAtt, demb, wgrad_ws, wgrad_ws_bytes, ln_part[2], embw_scratch;
  // dY operands of a layer's weight gradients (live from their producer to the layer's grouped weight-gradient launch)
  size_t gZ, gY, gZ1, gY1, gH, gQKV;
  size_t total;
};

__global__ void bf16_to_f32_kernel(const bf16* __restrict__ s, float* __restrict__ d, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(s)[i];
    f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    f32x4 b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    reinterpret_cast<f32x4*>(d)[2 * i] = a;
    reinterpret_cast<f32x4*>(d)[2 * i + 1] = b;
  }
}
// dst = (add ? dst : 0) + src
__global__ void f32_into_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, size_t n8, int add) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(s)[2 * i], b = reinterpret_cast<const f32x4*>(s)[2 * i + 1];
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (add) {
      const bf16x8 o = reinterpret_cast<const bf16x8*>(d)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += (float)o[e];
    }
    reinterpret_cast<bf16x8*>(d)[i] = pack8(f);
  }
}
__global__ void cast_pad_kernel(const float* __restrict__ s, bf16* __restrict__ d, int rows, int cols, int ld) {
  const long n = (long)rows * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), c = (int)(i % ld);
    d[i] = (bf16)(c < cols ? s[(long)r * cols + c] : 0.f);
  }
}
__global__ void unpad_kernel(const float* __restrict__ s, float* __restrict__ d, int rows, int cols, int ld, int add) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float v = s[(long)r * ld + c];
    d[i] = add ? d[i] + v : v;
  }
}
// every transposed weight shadow in ONE launch: table entry = {src float offset, dst byte offset, rows, cols, first tile}
struct TransDesc { unsigned long long src_off, dst_off; int rows, cols, tile0, tiles_x; };
__global__ __launch_bounds__(256) void transpose_table_kernel(const float* __restrict__ params, unsigned char* __restrict__ shadow,
                                                              const TransDesc* __restrict__ tab, int nent) {
  __shared__ float t[32][33];
  int e = 0;
  while (e + 1 < nent && (int)blockIdx.x >= tab[e + 1].tile0) ++e;
  const TransDesc d = tab[e];
  const int local = blockIdx.x - d.tile0;
  const int bx = (local % d.tiles_x) * 32, by = (local / d.tiles_x) * 32;
  const float* src = params + d.src_off;
  bf16* dst = reinterpret_cast<bf16*>(shadow + d.dst_off);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    t[i][tx] = (r < d.rows && c < d.cols) ? src[(long)r * d.cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (c < d.cols && r < d.rows) dst[(long)c * d.rows + r] = (bf16)t[tx][i];
  }
}

__global__ void set_u32_kernel(uint32_t* p, uint32_t v, int bump) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = bump ? *p + 1u : v;
}

inline int blocks_for(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

struct iq_model {
  iq_model_cfg_t c;
  int S, tok, has_cls, P, Ppad, dh, pool, head_ln;
  std::vector<Entry> entries;
  size_t nparam;
  size_t emb_w, emb_b, cls, emb_end;
  std::vector<LayerOff> L;
  size_t hln_g, hln_b, head_w, head_b, head_begin;
  size_t sh_flat, sh_embw, sh_table, shadow_bytes;
  std::vector<TransDesc> ttab;
  int ttiles = 0;
  const void* table_uploaded_to = nullptr;
  float* params = nullptr;
  float* grads = nullptr;
  const float* pe = nullptr;
  unsigned char* shadow = nullptr;
  std::string err;
  uint32_t* step_ctr = nullptr;   // caller-owned persistent device counter of the dropout step (iq_model_bind_step_counter)
  uint64_t last_seed = 0;   // seed of the last training forward; backward regenerates the same masks
  bool last_tr = false;     // whether the last forward applied dropout
  bool last_train_fwd = false;   // a forward has run in this workspace (the one-launch feed-forward's gate bits exist)

  iq_dropout_t bwd_site(uint32_t id, const uint32_t* step_dev, bool tr) const {
    iq_dropout_t d;
    d.seed = last_seed; d.step = 0; d.site = id; d.p = tr ? c.drop_prob : 0.f; d.step_dev = step_dev;
    return d;
  }

  size_t add(const std::string& name, std::vector<int> dims, size_t& cur, bool aligned = true) {
    if (aligned) cur = align_up(cur, 64);
    Entry e;
    e.name = name; e.off = cur; e.ndim = (int)dims.size();
    size_t n = 1;
    for (int i = 0; i < 4; ++i) { e.dims[i] = i < e.ndim ? dims[i] : 1; n *= (size_t)e.dims[i]; }
    entries.push_back(e);
    cur += n;
    return e.off;
  }
  const bf16* sh(size_t off) const { return reinterpret_cast<const bf16*>(shadow + sh_flat) + off; }
  const bf16* sht(size_t byte_off) const { return reinterpret_cast<const bf16*>(shadow + byte_off); }
};

namespace {

int fail(iq_model* m, int code, const std::string& msg) {
  if (m) m->err = msg;
  return code;
}

WsPlan plan_ws(const iq_model* m, int B) {
  WsPlan w;
  const size_t D = m->c.d_model, F = m->c.ffn_hidden, S = m->S, H = m->c.n_head;
  const size_t M = (size_t)B * S, MT = (size_t)B * m->tok;
  size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
  w.step_ctr = take(256);
  w.patches = take(MT * m->Ppad * 2);
  w.x0 = take(M * D * 2);
  w.head_feat = take((size_t)B * D * 4);
  w.head_stat = take((size_t)B * 2 * 4);
  w.layers.resize(m->c.n_layers);
  for (auto& l : w.layers) {
    l.qkv = take(M * 3 * D * 2);
    l.att = take(M * D * 2);
    l.lse = take((size_t)B * H * S * 4);
    l.z1 = take(M * D * 2);
    l.mean1 = take(M * 4);
    l.rstd1 = take(M * 4);
    l.x1 = take(M * D * 2);
    l.hid = take(M * F * 2);
    l.gate = take(iq_ffn_chain_gate_bytes((int)M, (int)F));     // "hid > 0" bits of the one-launch feed-forward (forward -> backward)
    l.z2 = take(M * D * 2);
    l.mean2 = take(M * 4);
    l.rstd2 = take(M * 4);
    l.x2 = take(M * D * 2);
  }
  w.gA = take(M * D * 2);
  w.gB = take(M * D * 2);
  w.gZ = take(M * D * 2);
  w.gY = take(M * D * 2);
  w.gZ1 = take(M * D * 2);
  w.gY1 = take(M * D * 2);
  w.gH = take(M * F * 2);
  w.gQKV = take(M * 3 * D * 2);
  w.gAtt = take(M * D * 2);
  w.demb = take(MT * D * 2);
  size_t wb = 0;
  auto mx = [&](size_t v) { if (v > wb) wb = v; };
  {
    iq_wgrad_problem_t g[4];
    memset(g, 0, sizeof(g));
    g[0].N = (int)D; g[0].K = (int)F;
    g[1].N = (int)F; g[1].K = (int)D;
    g[2].N = (int)D; g[2].K = (int)D;
    g[3].N = (int)(3 * D); g[3].K = (int)D;
    mx(iq_wgrad_grouped_ws_bytes(g, 4, (int)M, 0));
    for (int k = 0; k < 4; ++k) mx(iq_wgrad_grouped_ws_bytes(g + k, 1, (int)M, 0));
  }
  mx(iq_wgrad_ws_bytes((int)MT, (int)D, m->Ppad));
  w.wgrad_ws_bytes = wb;
  w.wgrad_ws = take(wb);
  // norm2 / norm1 partial rows, reduced with the layer's slabs: the stand-alone kernel caps its grid (one row per
  // block), the fused data-gradient GEMM + LayerNorm backward writes one row per row block of M
  size_t ln_rows_fused = (size_t)iq_gemm_lnbwd_partial_rows((int)M);
  if ((size_t)iq_ffn_chain_bwd_partial_rows((int)M) > ln_rows_fused) ln_rows_fused = (size_t)iq_ffn_chain_bwd_partial_rows((int)M);
  ln_rows_fused *= 2 * D * sizeof(float);
  const size_t ln_bytes = ln_rows_fused > iq_ln_bwd_ws_bytes((int)D) ? ln_rows_fused : iq_ln_bwd_ws_bytes((int)D);
  for (int k = 0; k < 2; ++k) w.ln_part[k] = take(ln_bytes);
  w.embw_scratch = take((size_t)D * m->Ppad * 4 + 256);
  w.total = cur;
  return w;
}

// The fused GEMM + LayerNorm launch where it was measured faster than GEMM then LayerNorm (scripts/gemm_bench.py ... ln,
// M = 50432: D 192 K 192 22.4 vs 25.9 us, K 768 38.9 vs 46.0; D 128 K 128 16.3 vs 20.6, K 1024 31.4 vs 34.4; D 256 K 256 31.6 vs
// 33.1 but K 1024 62.3 vs 55.7: its 64-row blocks re-stream the 512 KB weight twice as often).
inline bool use_fused_ln(int D, int K) { return iq_gemm_ln_supported(D, K) && (D <= 192 || K <= 256); }

// The encoder layer from the attention output on as one launch per direction (ffn_chain.hip).  IQ_TUNE_FFN_CHAIN=0|1 forces the
// choice (probes); IQ_TUNE_FFN_CHAIN_MIN_ROWS moves the threshold.  Measured (same box each, profiles/r03_probes.txt): cfg B
// (M = 50,432 rows, D 192, F 768, 7 waves of 32 rows per workgroup) 5.72 -> 4.88 ms per step; cfg C (M = 16,640, D 128, F 1024,
// 5 waves of 16 rows) 1.313 -> 1.191 ms; cfg A (M = 1,280: 16 workgroups) equal -- below 8,192 rows the tiled GEMMs stay.
inline bool use_ffn_chain(int M, int S, int D, int F) {
  static const int tune = [] { const char* e = getenv("IQ_TUNE_FFN_CHAIN"); return e ? atoi(e) : -1; }();
  static const int min_rows = [] { const char* e = getenv("IQ_TUNE_FFN_CHAIN_MIN_ROWS"); return e ? atoi(e) : 8192; }();
  if (!iq_ffn_chain_supported(S, D, F) || tune == 0) return false;
  if (tune == 1) return true;
  return M >= min_rows;
}

// ... with the attention output projection + norm1 as its first stage (IQ_TUNE_CHAIN_PRE=0: the projection stays a launch of its
// own) and the next layer's q,k,v projection as its last, the backward launch with the output projection's data gradient behind
// it (2; 1: without).  3: the backward launch also takes the q,k,v data gradient of the layer above + norm2 backward in front
// (iq_qkv_dgrad_ffn_chain_bwd).  Measured: cfg B 108.5 us against 68.1 + 39.6, step 4.896 vs 4.876 ms -- that stage moves 134 MB
// for 11 GFLOP and every workgroup runs it at the same time, it is as HBM-bound inside the launch as outside; on the small
// geometries a launch less is worth 1 % (cfg C 1.190 -> 1.180 ms, reference default ViT 1.629 -> 1.612).  Default: 3 up to
// 32,768 rows, 2 above.
inline int use_chain_pre(int M) {
  static const int tune = [] { const char* e = getenv("IQ_TUNE_CHAIN_PRE"); return e ? atoi(e) : -1; }();
  return tune >= 0 ? tune : M <= 32768 ? 3 : 2;
}

#define IQ_TRY(expr, what)                                                              \
  do {                                                                                  \
    int rc_ = (expr);                                                                   \
    if (rc_ != IQ_OK) return fail(m, rc_, std::string(what) + " failed (status " + std::to_string(rc_) + ")"); \
  } while (0)

iq_dropout_t site(const iq_model* m, uint64_t seed, const uint32_t* step_dev, uint32_t id, bool training) {
  iq_dropout_t d;
  d.seed = seed; d.step = 0; d.site = id; d.p = training ? m->c.drop_prob : 0.f; d.step_dev = step_dev;
  return d;
}

}  // namespace

extern "C" int iq_model_create(const iq_model_cfg_t* cfg, iq_model_t** out) {
  if (!cfg || !out) return IQ_ERR_ARG;
  *out = nullptr;
  iq_model* m = new iq_model();
  m->c = *cfg;
  const iq_model_cfg_t& c = m->c;
  auto bad = [&](const char* msg) { fprintf(stderr, "iq_model_create: %s\n", msg); delete m; return IQ_ERR_UNSUPPORTED; };
  if (c.d_model <= 0 || c.n_head <= 0 || c.n_layers < 0 || c.ffn_hidden <= 0 || c.num_classes <= 0 || c.in_channels <= 0)
    { delete m; return IQ_ERR_ARG; }
  if (c.d_model % c.n_head) return bad("d_model must be divisible by n_head");
  m->dh = c.d_model / c.n_head;
  if (m->dh != 16 && m->dh != 32 && m->dh != 64) return bad("head dim (d_model/n_head) must be 16, 32 or 64");
  if (c.ffn_hidden % 8) return bad("ffn_hidden must be a multiple of 8");
  if (!iq_ln_supported(c.d_model)) return bad("d_model not supported by the LayerNorm kernel");
  if (!(c.drop_prob >= 0.f && c.drop_prob < 1.f)) return bad("drop_prob must be in [0,1)");
  if (c.kind == 0) {
    if (c.patch <= 0 || c.img_h < c.patch || c.img_w < c.patch) return bad("bad image / patch size");
    m->tok = (c.img_h / c.patch) * (c.img_w / c.patch);
    m->has_cls = 1;
    m->P = c.in_channels * c.patch * c.patch;
    m->pool = 0;
    m->head_ln = 0;
  } else if (c.kind == 1) {
    if (c.conv_k <= 0 || c.seq_length < c.conv_k || (c.seq_length % c.conv_k)) return bad("bad seq_length / segment_size");
    m->tok = c.seq_length / c.conv_k;
    m->has_cls = c.use_cls ? 1 : 0;
    m->P = c.in_channels * c.conv_k;
    m->pool = m->has_cls ? 0 : 1;
    m->head_ln = 1;
  } else {
    delete m;
    return IQ_ERR_ARG;
  }
  m->S = m->tok + m->has_cls;
  m->Ppad = (int)align_up((size_t)m->P, 32);
  if (!iq_attn_supported(m->S, m->dh)) return bad("sequence too long for the attention backward kernel at this head dim");

  const int D = c.d_model, F = c.ffn_hidden, K = c.num_classes;
  size_t cur = 0;
  const std::string emb = c.kind == 0 ? "encoder.patch_embedding.projection" : "encoder.sequence_embedding.projection";
  if (c.kind == 0) m->emb_w = m->add(emb + ".weight", {D, c.in_channels, c.patch, c.patch}, cur);
  else m->emb_w = m->add(emb + ".weight", {D, c.in_channels, c.conv_k}, cur);
  m->emb_b = m->add(emb + ".bias", {D}, cur);
  m->cls = 0;
  if (m->has_cls) m->cls = m->add("encoder.cls_token", {1, 1, D}, cur);
  cur = align_up(cur, 64);
  m->emb_end = cur;
  m->L.resize(c.n_layers);
  for (int i = 0; i < c.n_layers; ++i) {
    LayerOff& o = m->L[i];
    const std::string p = "encoder.layers." + std::to_string(i) + ".";
    o.wqkv = m->add(p + "attention.w_q.weight", {D, D}, cur);
    m->add(p + "attention.w_k.weight", {D, D}, cur, false);
    m->add(p + "attention.w_v.weight", {D, D}, cur, false);
    o.bqkv = m->add(p + "attention.w_q.bias", {D}, cur);
    m->add(p + "attention.w_k.bias", {D}, cur, false);
    m->add(p + "attention.w_v.bias", {D}, cur, false);
    o.wo = m->add(p + "attention.w_concat.weight", {D, D}, cur);
    o.bo = m->add(p + "attention.w_concat.bias", {D}, cur);
    o.g1 = m->add(p + "norm1.gamma", {D}, cur);
    o.be1 = m->add(p + "norm1.beta", {D}, cur);
    o.w1 = m->add(p + "ffn.linear1.weight", {F, D}, cur);
    o.b1 = m->add(p + "ffn.linear1.bias", {F}, cur);
    o.w2 = m->add(p + "ffn.linear2.weight", {D, F}, cur);
    o.b2 = m->add(p + "ffn.linear2.bias", {D}, cur);
    o.g2 = m->add(p + "norm2.gamma", {D}, cur);
    o.be2 = m->add(p + "norm2.beta", {D}, cur);
    cur = align_up(cur, 64);
    o.end = cur;
  }
  m->head_begin = cur;
  m->hln_g = m->hln_b = 0;
  if (m->head_ln) {
    m->hln_g = m->add("mlp_head.0.weight", {D}, cur);
    m->hln_b = m->add("mlp_head.0.bias", {D}, cur);
    m->head_w = m->add("mlp_head.1.weight", {K, D}, cur);
    m->head_b = m->add("mlp_head.1.bias", {K}, cur);
  } else {
    m->head_w = m->add("mlp_head.weight", {K, D}, cur);
    m->head_b = m->add("mlp_head.bias", {K}, cur);
  }
  m->nparam = align_up(cur, 64);

  // shadow layout
  size_t sb = 0;
  m->sh_flat = sb; sb = align_up(sb + m->nparam * 2, 256);
  m->sh_embw = sb; sb = align_up(sb + (size_t)D * m->Ppad * 2, 256);
  for (auto& o : m->L) {
    o.t_wqkv = sb; sb = align_up(sb + (size_t)3 * D * D * 2, 256);
    o.t_wo = sb; sb = align_up(sb + (size_t)D * D * 2, 256);
    o.t_w1 = sb; sb = align_up(sb + (size_t)F * D * 2, 256);
    o.t_w2 = sb; sb = align_up(sb + (size_t)D * F * 2, 256);
  }
  for (auto& o : m->L) {
    auto push = [&](size_t src, size_t dst, int rows, int cols) {
      TransDesc d;
      d.src_off = src; d.dst_off = dst; d.rows = rows; d.cols = cols; d.tile0 = m->ttiles; d.tiles_x = (cols + 31) / 32;
      m->ttiles += d.tiles_x * ((rows + 31) / 32);
      m->ttab.push_back(d);
    };
    push(o.wqkv, o.t_wqkv, 3 * D, D);
    push(o.wo, o.t_wo, D, D);
    push(o.w1, o.t_w1, F, D);
    push(o.w2, o.t_w2, D, F);
  }
  m->sh_table = sb; sb = align_up(sb + m->ttab.size() * sizeof(TransDesc) + 64, 256);
  m->shadow_bytes = sb;
  *out = m;
  return IQ_OK;
}

extern "C" void iq_model_destroy(iq_model_t* m) {
  if (!m) return;
  delete m;
}
extern "C" const char* iq_model_last_error(const iq_model_t* m) { return m ? m->err.c_str() : "null model"; }
extern "C" int iq_model_tokens(const iq_model_t* m) { return m ? m->S : 0; }
extern "C" size_t iq_model_param_floats(const iq_model_t* m) { return m ? m->nparam : 0; }
extern "C" int iq_model_param_entries(const iq_model_t* m) { return m ? (int)m->entries.size() : 0; }
extern "C" int iq_model_param_entry(const iq_model_t* m, int i, char* name, int name_cap, size_t* offset, int* ndim,
                                    int* dims) {
  if (!m || i < 0 || i >= (int)m->entries.size() || !name || name_cap <= 0 || !offset || !ndim || !dims) return IQ_ERR_ARG;
  const Entry& e = m->entries[i];
  snprintf(name, (size_t)name_cap, "%s", e.name.c_str());
  *offset = e.off;
  *ndim = e.ndim;
  for (int k = 0; k < 4; ++k) dims[k] = e.dims[k];
  return IQ_OK;
}
extern "C" size_t iq_model_shadow_bytes(const iq_model_t* m) { return m ? m->shadow_bytes : 0; }
extern "C" size_t iq_model_workspace_bytes(const iq_model_t* m, int batch, int training) {
  (void)training;
  if (!m || batch <= 0) return 0;
  return plan_ws(m, batch).total;
}

extern "C" int iq_model_bind(iq_model_t* m, float* params, float* grads, const float* pe, void* shadow) {
  if (!m) return IQ_ERR_ARG;
  if (!params || !pe || !shadow) return fail(m, IQ_ERR_ARG, "bind: params, pe and shadow are required");
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)shadow) & 255) return fail(m, IQ_ERR_ARG, "bind: buffers must be 256 B aligned");
  m->params = params; m->grads = grads; m->pe = pe; m->shadow = (unsigned char*)shadow;
  if (m->table_uploaded_to != shadow && !m->ttab.empty()) {
    // one small synchronous copy per (re)binding of the shadow buffer; never on the step path
    if (hipMemcpy(m->shadow + m->sh_table, m->ttab.data(), m->ttab.size() * sizeof(TransDesc), hipMemcpyHostToDevice) != hipSuccess)
      return fail(m, IQ_ERR_LAUNCH, "bind: uploading the transpose table failed");
    m->table_uploaded_to = shadow;
  }
  return IQ_OK;
}

extern "C" int iq_model_bind_step_counter(iq_model_t* m, uint32_t* counter) {
  if (!m) return IQ_ERR_ARG;
  if ((uintptr_t)counter & 3) return fail(m, IQ_ERR_ARG, "bind_step_counter: counter must be 4 B aligned");
  m->step_ctr = counter;
  return IQ_OK;
}

static int refresh_impl(iq_model_t* m, bool cast_flat, iq_stream_t stream) {
  if (!m || !m->params || !m->shadow) return fail(m, IQ_ERR_ARG, "refresh_shadow: model not bound");
  hipStream_t st = (hipStream_t)stream;
  const int D = m->c.d_model;
  if (cast_flat) IQ_TRY(iq_cast_bf16(m->params, m->shadow + m->sh_flat, m->nparam, stream), "cast params");
  IQ_PROF(IQ_FAM_MISC, st);
  cast_pad_kernel<<<blocks_for((size_t)D * m->Ppad), 256, 0, st>>>(m->params + m->emb_w, (bf16*)(m->shadow + m->sh_embw), D,
                                                                  m->P, m->Ppad);
  if (m->ttiles > 0)
    transpose_table_kernel<<<m->ttiles, 256, 0, st>>>(m->params, m->shadow, (const TransDesc*)(m->shadow + m->sh_table),
                                                     (int)m->ttab.size());
  return iq_launch_status();
}

extern "C" int iq_model_refresh_shadow(iq_model_t* m, iq_stream_t stream) { return refresh_impl(m, true, stream); }
// after iq_adamw_step has already written the flat bf16 mirror: only the transposed / padded copies
extern "C" int iq_model_refresh_transposed(iq_model_t* m, iq_stream_t stream) { return refresh_impl(m, false, stream); }

extern "C" int iq_model_forward(iq_model_t* m, const float* src, int batch, void* workspace, size_t ws_bytes,
                                int training, uint64_t seed, uint32_t step, float* enc_out, float* logits,
                                iq_stream_t stream) {
  if (!m) return IQ_ERR_ARG;
  if (!m->params || !m->shadow || !m->pe) return fail(m, IQ_ERR_ARG, "forward: model not bound");
  if (!src || !workspace || batch <= 0) return fail(m, IQ_ERR_ARG, "forward: bad arguments");
  if ((uintptr_t)workspace & 255) return fail(m, IQ_ERR_ARG, "forward: workspace must be 256 B aligned");
  const WsPlan w = plan_ws(m, batch);
  if (ws_bytes < w.total) return fail(m, IQ_ERR_ARG, "forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  const iq_model_cfg_t& c = m->c;
  const int D = c.d_model, F = c.ffn_hidden, S = m->S, H = c.n_head, B = batch;
  const int M = B * S, MT = B * m->tok;
  const bool tr = training != 0 && c.drop_prob > 0.f;
  // dropout step: the caller's persistent counter when bound (survives workspace reallocation, starts from a defined
  // value), else a slot of the workspace
  uint32_t* step_dev = m->step_ctr ? m->step_ctr : (uint32_t*)(ws + w.step_ctr);
  m->last_tr = tr;
  m->last_train_fwd = true;      // (every forward leaves the gate bits of the one-launch feed-forward where it runs)
  if (tr) {
    set_u32_kernel<<<1, 64, 0, st>>>(step_dev, step, step == 0xFFFFFFFFu ? 1 : 0);
    m->last_seed = seed;
  }
  const float* P = m->params;

  // ---- embedding: patchify -> GEMM(+bias +PE, dropout) ; cls rows -------------------------------
  if (c.kind == 0) IQ_TRY(iq_patchify(src, ws + w.patches, 0, B, c.in_channels, c.img_h, c.img_w, c.patch, m->Ppad, stream), "patchify");
  else IQ_TRY(iq_patchify(src, ws + w.patches, 1, B, c.in_channels, c.seq_length, 0, c.conv_k, m->Ppad, stream), "patchify");
  const iq_dropout_t d0 = site(m, seed, step_dev, 0, tr);
  if (m->has_cls) IQ_TRY(iq_cls_rows(P + m->cls, m->pe, ws + w.x0, B, S, D, &d0, stream), "cls rows");
  {
    iq_epilogue_t e;
    memset(&e, 0, sizeof(e));
    e.bias = P + m->emb_b; e.pe = m->pe; e.tok = m->tok; e.seq = S; e.cls_off = m->has_cls; e.drop = d0;
    IQ_TRY(iq_gemm_bf16_nt(ws + w.patches, m->Ppad, m->shadow + m->sh_embw, m->Ppad, ws + w.x0, D, MT, D, m->Ppad, &e, stream),
           "embedding GEMM");
  }
  const unsigned char* x = ws + w.x0;
  bool qkv_done = false;
  for (int l = 0; l < c.n_layers; ++l) {
    const LayerOff& o = m->L[l];
    const WsPlan::L& a = w.layers[l];
    iq_epilogue_t e;
    // q,k,v projections as one [3D x D] GEMM over the packed weight view
    // (not when the previous layer's last launch has already written it: iq_attn_out_ffn_chain_fwd's q,k,v stage)
    if (!qkv_done) {
      memset(&e, 0, sizeof(e));
      e.bias = P + o.bqkv;
      IQ_TRY(iq_gemm_bf16_nt(x, D, m->sh(o.wqkv), D, ws + a.qkv, 3 * D, M, 3 * D, D, &e, stream), "qkv GEMM");
    }
    qkv_done = false;
    IQ_TRY(iq_attn_fwd(ws + a.qkv, ws + a.att, (float*)(ws + a.lse), B, S, H, m->dh, stream), "attention fwd");
    // out-proj + dropout1 + residual + norm1: one launch when a workgroup can own whole rows (D <= 256), else two -- or, where
    // the FFN chain kernel runs, its first stage (iq_attn_out_ffn_chain_fwd: the layer from the attention output on is one launch)
    const iq_dropout_t dr1 = site(m, seed, step_dev, 1 + 3 * l, tr);
    const iq_dropout_t drh = site(m, seed, step_dev, 2 + 3 * l, tr);
    const iq_dropout_t dr2 = site(m, seed, step_dev, 3 + 3 * l, tr);
    if (use_ffn_chain(M, S, D, F) && use_chain_pre(M)) {
      const bool next = l + 1 < c.n_layers && use_chain_pre(M) >= 2;
      IQ_TRY(iq_attn_out_ffn_chain_fwd(ws + a.att, m->sh(o.wo), P + o.bo, &dr1, x, P + o.g1, P + o.be1, ws + a.z1, ws + a.x1,
                                       (float*)(ws + a.mean1), (float*)(ws + a.rstd1), m->sh(o.w1), P + o.b1, &drh, ws + a.hid,
                                       m->sh(o.w2), P + o.b2, &dr2, P + o.g2, P + o.be2, 1e-12f, ws + a.z2, ws + a.x2,
                                       (float*)(ws + a.mean2), (float*)(ws + a.rstd2), ws + a.gate,
                                       next ? m->sh(m->L[l + 1].wqkv) : nullptr, next ? P + m->L[l + 1].bqkv : nullptr,
                                       next ? ws + w.layers[l + 1].qkv : nullptr, B, S, D, F, stream),
             "out-proj + norm1 + ffn chain + norm2 (+ next q,k,v)");
      qkv_done = next;
      x = ws + a.x2;
      continue;
    }
    if (use_fused_ln(D, D)) {
      IQ_TRY(iq_gemm_bf16_ln(ws + a.att, D, m->sh(o.wo), D, P + o.bo, x, D, &dr1, P + o.g1, P + o.be1, 1e-12f, ws + a.z1,
                             ws + a.x1, (float*)(ws + a.mean1), (float*)(ws + a.rstd1), M, D, D, stream), "out-proj GEMM + norm1");
    } else {
      memset(&e, 0, sizeof(e));
      e.bias = P + o.bo; e.drop = dr1; e.residual = x; e.ldr = D;
      IQ_TRY(iq_gemm_bf16_nt(ws + a.att, D, m->sh(o.wo), D, ws + a.z1, D, M, D, D, &e, stream), "out-proj GEMM");
      IQ_TRY(iq_ln_fwd(ws + a.z1, P + o.g1, P + o.be1, ws + a.x1, (float*)(ws + a.mean1), (float*)(ws + a.rstd1), M, D, 1e-12f, stream), "norm1");
    }
    // ffn + dropout2 + residual + norm2: one launch per layer where a frame's rows fit one workgroup (iq_ffn_chain_fwd: the
    // hidden activation is consumed from LDS), else FFN1 then FFN2 (+ norm2 in its epilogue where a workgroup owns whole rows)
    if (use_ffn_chain(M, S, D, F)) {
      IQ_TRY(iq_ffn_chain_fwd(ws + a.x1, m->sh(o.w1), P + o.b1, &drh, ws + a.hid, m->sh(o.w2), P + o.b2, &dr2, P + o.g2, P + o.be2,
                              1e-12f, ws + a.z2, ws + a.x2, (float*)(ws + a.mean2), (float*)(ws + a.rstd2), ws + a.gate, B, S, D, F, stream),
             "ffn chain + norm2");
    } else {
      memset(&e, 0, sizeof(e));
      e.bias = P + o.b1; e.relu = 1; e.drop = drh;
      IQ_TRY(iq_gemm_bf16_nt(ws + a.x1, D, m->sh(o.w1), D, ws + a.hid, F, M, F, D, &e, stream), "ffn1 GEMM");
      if (use_fused_ln(D, F)) {
        IQ_TRY(iq_gemm_bf16_ln(ws + a.hid, F, m->sh(o.w2), F, P + o.b2, ws + a.x1, D, &dr2, P + o.g2, P + o.be2, 1e-12f,
                               ws + a.z2, ws + a.x2, (float*)(ws + a.mean2), (float*)(ws + a.rstd2), M, D, F, stream), "ffn2 GEMM + norm2");
      } else {
        memset(&e, 0, sizeof(e));
        e.bias = P + o.b2; e.drop = dr2; e.residual = ws + a.x1; e.ldr = D;
        IQ_TRY(iq_gemm_bf16_nt(ws + a.hid, F, m->sh(o.w2), F, ws + a.z2, D, M, D, F, &e, stream), "ffn2 GEMM");
        IQ_TRY(iq_ln_fwd(ws + a.z2, P + o.g2, P + o.be2, ws + a.x2, (float*)(ws + a.mean2), (float*)(ws + a.rstd2), M, D, 1e-12f, stream), "norm2");
      }
    }
    x = ws + a.x2;
  }
  if (logits) {
    IQ_TRY(iq_head_fwd(x, m->head_ln ? P + m->hln_g : nullptr, m->head_ln ? P + m->hln_b : nullptr, P + m->head_w,
                       P + m->head_b, (float*)(ws + w.head_feat), (float*)(ws + w.head_stat), logits, B, S, D,
                       c.num_classes, m->pool, stream), "head");
  }
  if (enc_out) bf16_to_f32_kernel<<<blocks_for((size_t)M * D / 8), 256, 0, st>>>((const bf16*)x, enc_out, (size_t)M * D / 8);
  return iq_launch_status();
}

extern "C" int iq_model_grad_range(const iq_model_t* m, int stage_hi, int stage_lo, size_t* off, size_t* len) {
  if (!m || !off || !len) return IQ_ERR_ARG;
  const int Lr = m->c.n_layers;
  if (stage_lo < 0 || stage_hi > Lr + 1 || stage_lo > stage_hi) return IQ_ERR_ARG;
  auto begin = [&](int s) { return s == 0 ? (size_t)0 : (s <= Lr ? (s == 1 ? m->emb_end : m->L[s - 2].end) : m->head_begin); };
  auto end = [&](int s) { return s == 0 ? m->emb_end : (s <= Lr ? m->L[s - 1].end : m->nparam); };
  *off = begin(stage_lo);
  *len = end(stage_hi) - *off;
  return IQ_OK;
}

extern "C" int iq_model_backward(iq_model_t* m, const float* dlogits, const float* denc, int batch, void* workspace,
                                 size_t ws_bytes, int accumulate, int stage_hi, int stage_lo, iq_stream_t stream) {
  if (!m) return IQ_ERR_ARG;
  if (!m->params || !m->grads || !m->shadow) return fail(m, IQ_ERR_ARG, "backward: model not bound (grads required)");
  if (!workspace || batch <= 0) return fail(m, IQ_ERR_ARG, "backward: bad arguments");
  const iq_model_cfg_t& c = m->c;
  const int Lr = c.n_layers;
  if (stage_lo < 0 || stage_hi > Lr + 1 || stage_lo > stage_hi) return fail(m, IQ_ERR_ARG, "backward: bad stage range");
  const WsPlan w = plan_ws(m, batch);
  if (ws_bytes < w.total) return fail(m, IQ_ERR_ARG, "backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  const int D = c.d_model, F = c.ffn_hidden, S = m->S, H = c.n_head, B = batch, K = c.num_classes;
  const int M = B * S, MT = B * m->tok;
  const bool tr = m->last_tr;   // masks are regenerated only if the forward in this workspace applied them
  const uint32_t* step_dev = m->step_ctr ? m->step_ctr : (const uint32_t*)(ws + w.step_ctr);
  const float* P = m->params;
  float* G = m->grads;
  float* wws = (float*)(ws + w.wgrad_ws);
  const float dscale = tr ? dropout_scale(c.drop_prob) : 1.f;

  if (stage_hi == Lr + 1) {
    if (!dlogits && !denc) return fail(m, IQ_ERR_ARG, "backward: dlogits or denc required at the head stage");
    if (dlogits) {
      IQ_TRY(iq_head_bwd(dlogits, (const float*)(ws + w.head_feat), (const float*)(ws + w.head_stat),
                         m->head_ln ? P + m->hln_g : nullptr, m->head_ln ? P + m->hln_b : nullptr, P + m->head_w,
                         G + m->head_w, G + m->head_b, m->head_ln ? G + m->hln_g : nullptr,
                         m->head_ln ? G + m->hln_b : nullptr, ws + w.gA, B, S, D, K, m->pool, accumulate, stream), "head bwd");
    } else if (!accumulate) {
      (void)hipMemsetAsync(G + m->head_begin, 0, (m->nparam - m->head_begin) * sizeof(float), st);
    }
    if (denc) f32_into_bf16_kernel<<<blocks_for((size_t)M * D / 8), 256, 0, st>>>(denc, (bf16*)(ws + w.gA), (size_t)M * D / 8, dlogits ? 1 : 0);
  }
  // The four weight gradients of a layer are independent of its dX chain and of each other, and their dY operands
  // (gY|gZ, gH, gY1|gZ1, gQKV) all stay live until the end of the layer: they run as ONE grouped launch (+ one slab
  // reduce) instead of four (+ four) -- one pipeline fill / drain, 3x fewer slab bytes.  (Measured and not kept, see
  // profiles/r01_probes.txt: the launch on a side stream with a workgroup budget next to the following layer's chain
  // 6.57-7.29 vs 6.58 ms/step; each weight gradient right after its data-gradient GEMM 6.35 vs 6.15-6.22; the FFN
  // data-gradient pair as one chained launch 6.65-6.68 vs 6.62-6.64.)
  // LayerNorm backward rides in the epilogue of the GEMM that produces its incoming gradient where a workgroup owns
  // whole rows (D = 128 | 192): norm1's in the FFN1 data gradient of the same layer, norm2's in the QKV data gradient of
  // the layer ABOVE (the top layer's comes from the head and keeps the stand-alone kernel).
  const bool fuse1 = iq_gemm_lnbwd_supported(D, F) != 0, fuse2 = iq_gemm_lnbwd_supported(D, 3 * D) != 0;
  bool deferred = false;      // the layer above left its q,k,v data gradient + this layer's norm2 backward to this layer's chain launch
  for (int sidx = (stage_hi > Lr ? Lr : stage_hi); sidx >= 1 && sidx >= stage_lo; --sidx) {
    const int l = sidx - 1;
    const LayerOff& o = m->L[l];
    const WsPlan::L& a = w.layers[l];
    const unsigned char* xin = l == 0 ? ws + w.x0 : ws + w.layers[l - 1].x2;
    unsigned char *gZ = ws + w.gZ, *gY = ws + w.gY, *gZ1 = ws + w.gZ1, *gY1 = ws + w.gY1;
    unsigned char *gH = ws + w.gH, *gQKV = ws + w.gQKV;
    iq_epilogue_t e;
    float* lp2 = (float*)(ws + w.ln_part[0]);
    float* lp1 = (float*)(ws + w.ln_part[1]);
    // norm2 backward (+ regenerated dropout2 mask) -- unless the layer above already did it in its QKV data gradient
    const bool norm2_here = l == Lr - 1 || !fuse2;
    if (norm2_here) {
      const iq_dropout_t dr2 = m->bwd_site(3 + 3 * l, step_dev, tr);
      IQ_TRY(iq_ln_bwd(ws + w.gA, ws + a.z2, (const float*)(ws + a.mean2), (const float*)(ws + a.rstd2), P + o.g2,
                       gZ, gY, &dr2, nullptr, nullptr, lp2, accumulate, M, D, stream), "norm2 bwd");
    }
    const unsigned char* dO2 = tr ? gY : gZ;
    const unsigned char* dAo = tr ? gY1 : gZ1;
    const iq_wgrad_problem_t wg[4] = {
        {dO2, D, ws + a.hid, F, G + o.w2, G + o.b2, D, F},            // ffn.linear2
        {gH, F, ws + a.x1, D, G + o.w1, G + o.b1, F, D},              // ffn.linear1
        {dAo, D, ws + a.att, D, G + o.wo, G + o.bo, D, D},            // attention.w_concat
        {gQKV, 3 * D, xin, D, G + o.wqkv, G + o.bqkv, 3 * D, D}};     // attention.w_q|w_k|w_v
    // the LayerNorm gamma/beta partial rows of this layer ride on the same reduce launch
    const int rows2 = norm2_here ? iq_ln_bwd_partial_rows(M, D) : deferred ? iq_ffn_chain_bwd_partial_rows(M) : iq_gemm_lnbwd_partial_rows(M);
    // the one-launch feed-forward backward (ffn_chain.hip) where the forward ran its one-launch counterpart (it left the gate bits)
    const bool chain = fuse1 && m->last_train_fwd && use_ffn_chain(M, S, D, F);
    const int rows1 = chain ? iq_ffn_chain_bwd_partial_rows(M) : fuse1 ? iq_gemm_lnbwd_partial_rows(M) : iq_ln_bwd_partial_rows(M, D);
    const iq_reduce_seg_t lnseg[4] = {{lp2, rows2, 2L * D, G + o.g2, D}, {lp2 + D, rows2, 2L * D, G + o.be2, D},
                                      {lp1, rows1, 2L * D, G + o.g1, D}, {lp1 + D, rows1, 2L * D, G + o.be1, D}};
    const iq_dropout_t dr1 = m->bwd_site(1 + 3 * l, step_dev, tr);
    const bool post = chain && use_chain_pre(M) >= 2;       // ... and the output projection's data gradient behind it
    if (chain && deferred) {
      const LayerOff& ou = m->L[l + 1];
      const iq_dropout_t dr2b = m->bwd_site(3 + 3 * l, step_dev, tr);
      IQ_TRY(iq_qkv_dgrad_ffn_chain_bwd(gQKV, m->sht(ou.t_wqkv), gZ1, ws + a.z2, (const float*)(ws + a.mean2), (const float*)(ws + a.rstd2),
                                        P + o.g2, &dr2b, gZ, gY, lp2, m->sht(o.t_w2), ws + a.gate, dscale, gH, m->sht(o.t_w1), gZ, ws + a.z1,
                                        (const float*)(ws + a.mean1), (const float*)(ws + a.rstd1), P + o.g1, &dr1, gZ1, gY1, lp1,
                                        m->sht(o.t_wo), ws + w.gAtt, B, S, D, F, stream),
             "qkv dgrad of the layer above + norm2 bwd + ffn chain bwd + norm1 bwd + out-proj dgrad");
    } else if (chain) {
      IQ_TRY(iq_ffn_chain_bwd(dO2, m->sht(o.t_w2), ws + a.gate, dscale, gH, m->sht(o.t_w1), gZ, ws + a.z1, (const float*)(ws + a.mean1),
                              (const float*)(ws + a.rstd1), P + o.g1, &dr1, gZ1, gY1, lp1, post ? m->sht(o.t_wo) : nullptr,
                              post ? ws + w.gAtt : nullptr, B, S, D, F, stream), "ffn chain bwd + norm1 bwd (+ out-proj dgrad)");
    } else {
      memset(&e, 0, sizeof(e));
      e.gate = ws + a.hid; e.ldg = F; e.gate_scale = dscale;
      IQ_TRY(iq_gemm_bf16_nt(dO2, D, m->sht(o.t_w2), D, gH, F, M, F, D, &e, stream), "ffn2 dgrad");
      // FFN1 data gradient (+ the residual-path gradient gZ) and norm1 backward (+ dropout1 mask)
      if (fuse1) {
        IQ_TRY(iq_gemm_bf16_lnbwd(gH, F, m->sht(o.t_w1), F, gZ, D, ws + a.z1, (const float*)(ws + a.mean1),
                                  (const float*)(ws + a.rstd1), P + o.g1, &dr1, gZ1, gY1, lp1, M, D, F, stream), "ffn1 dgrad + norm1 bwd");
      } else {
        memset(&e, 0, sizeof(e));
        e.residual = gZ; e.ldr = D;
        IQ_TRY(iq_gemm_bf16_nt(gH, F, m->sht(o.t_w1), F, ws + w.gB, D, M, D, F, &e, stream), "ffn1 dgrad");
        IQ_TRY(iq_ln_bwd(ws + w.gB, ws + a.z1, (const float*)(ws + a.mean1), (const float*)(ws + a.rstd1), P + o.g1,
                         gZ1, gY1, &dr1, nullptr, nullptr, lp1, accumulate, M, D, stream), "norm1 bwd");
      }
    }
    if (!post) IQ_TRY(iq_gemm_bf16_nt(dAo, D, m->sht(o.t_wo), D, ws + w.gAtt, D, M, D, D, nullptr, stream), "out-proj dgrad");
    IQ_TRY(iq_attn_bwd(ws + a.qkv, ws + a.att, ws + w.gAtt, (const float*)(ws + a.lse), gQKV, B, S, H, m->dh, stream), "attention bwd");
    // The four weight gradients of the layer, BEFORE the QKV data gradient: fused with the norm2 backward of the layer
    // below, that GEMM overwrites gZ / gY and the norm2 partial rows, which the weight gradients / their reduce still read.
    IQ_TRY(iq_gemm_bf16_wgrad_grouped(wg, 4, M, wws, w.wgrad_ws_bytes, accumulate, 0, lnseg, 4, stream), "layer weight gradients");
    // ... unless the layer below takes it into its own chain launch (iq_qkv_dgrad_ffn_chain_bwd)
    deferred = l > 0 && fuse2 && chain && post && use_chain_pre(M) >= 3 && sidx - 1 >= stage_lo;
    if (deferred) continue;
    if (l > 0 && fuse2) {
      const LayerOff& ob = m->L[l - 1];
      const WsPlan::L& ab = w.layers[l - 1];
      const iq_dropout_t dr2b = m->bwd_site(3 + 3 * (l - 1), step_dev, tr);
      IQ_TRY(iq_gemm_bf16_lnbwd(gQKV, 3 * D, m->sht(o.t_wqkv), 3 * D, gZ1, D, ws + ab.z2, (const float*)(ws + ab.mean2),
                                (const float*)(ws + ab.rstd2), P + ob.g2, &dr2b, gZ, gY, lp2, M, D, 3 * D, stream),
             "qkv dgrad + norm2 bwd of the layer below");
    } else {
      memset(&e, 0, sizeof(e));
      e.residual = gZ1; e.ldr = D;
      IQ_TRY(iq_gemm_bf16_nt(gQKV, 3 * D, m->sht(o.t_wqkv), 3 * D, ws + w.gA, D, M, D, 3 * D, &e, stream), "qkv dgrad");
    }
  }
  if (stage_lo == 0) {
    const iq_dropout_t dr0 = m->bwd_site(0, step_dev, tr);
    IQ_TRY(iq_embed_bwd_gather(ws + w.gA, ws + w.demb, m->has_cls ? G + m->cls : nullptr, B, S, m->tok, D, m->has_cls,
                               &dr0, accumulate, stream), "embedding bwd gather");
    if (m->Ppad == m->P) {
      IQ_TRY(iq_gemm_bf16_wgrad(ws + w.demb, D, ws + w.patches, m->Ppad, G + m->emb_w, G + m->emb_b, MT, D, m->Ppad, wws,
                                w.wgrad_ws_bytes, accumulate, stream), "embedding wgrad");
    } else {
      float* scratch = (float*)(ws + w.embw_scratch);
      if (accumulate) (void)hipMemsetAsync(scratch, 0, (size_t)D * m->Ppad * sizeof(float), st);
      IQ_TRY(iq_gemm_bf16_wgrad(ws + w.demb, D, ws + w.patches, m->Ppad, scratch, G + m->emb_b, MT, D, m->Ppad, wws,
                                w.wgrad_ws_bytes, accumulate, stream), "embedding wgrad");
      unpad_kernel<<<blocks_for((size_t)D * m->P), 256, 0, st>>>(scratch, G + m->emb_w, D, m->P, m->Ppad, accumulate);
    }
  }
  return iq_launch_status();
}
