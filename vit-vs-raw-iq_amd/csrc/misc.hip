// Small-FLOP kernels around the encoder (gfx950): embedding front end, classification head,
// label-smoothed cross entropy, gradient norm, fused clip+AdamW, bf16 shadow casts.
// All HBM/latency-bound; fp32 math; 16 B vectors where the data is wide enough to matter.
//
// Reference (under /root/reference/Transformer_Thesis/):
//   patchify ............ ViT/models/embedding/patch_embedding.py:11-15 (Conv2d k=s=p as reshape+GEMM)
//                         transformer_rawIQ/models/embedding/patch_embedding.py:47-60 (Conv1d k=s)
//   cls rows ............ ViT/models/encoder.py:42-47  (cls repeat + cat, + PE, dropout)
//   head ................ ViT/models/amc_transformer.py:29-30; transformer_rawIQ/models/transformer_rawIQ.py:88-96
//   loss ................ ViT/training/train.py:405 (CrossEntropyLoss(label_smoothing=0.1)), :205-207 (accuracy)
//   clip + AdamW ........ ViT/training/train.py:199-201,407-412
#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

// ---------------------------------------------------------------------------------------------
// embedding front end
// ---------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int kind, int B, int C, int H,
                                int W, int p, int tok, int P, int Kpad) {
  const long nchunk = (long)B * tok * (Kpad / 8);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < nchunk; id += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(id % (Kpad / 8));
    const long row = id / (Kpad / 8);
    const int b = (int)(row / tok), t = (int)(row % tok);
    float v[8];
    if (kind == 0 && (p & 7) == 0 && (W & 3) == 0 && ch * 8 < P) {
      // 8 consecutive px of one (c, py): 32 contiguous, 16 B-aligned source bytes
      const int k = ch * 8, gw = W / p;
      const int c = k / (p * p), rem = k % (p * p), py = rem / p, px = rem % p;
      const int gy = t / gw, gx = t % gw;
      const float* sp = src + (((long)b * C + c) * H + gy * p + py) * W + gx * p + px;
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp), bq = *reinterpret_cast<const f32x4*>(sp + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = bq[0]; v[5] = bq[1]; v[6] = bq[2]; v[7] = bq[3];
      *reinterpret_cast<bf16x8*>(dst + row * Kpad + ch * 8) = pack8(v);
      continue;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ch * 8 + e;
      float x = 0.f;
      if (k < P) {
        if (kind == 0) {
          const int gw = W / p;
          const int c = k / (p * p), rem = k % (p * p), py = rem / p, px = rem % p;
          const int gy = t / gw, gx = t % gw;
          x = src[(((long)b * C + c) * H + gy * p + py) * W + gx * p + px];
        } else {
          const int c = k / p, j = k % p;   // H = sequence length L, p = conv kernel
          x = src[((long)b * C + c) * H + (long)t * p + j];
        }
      }
      v[e] = x;
    }
    *reinterpret_cast<bf16x8*>(dst + row * Kpad + ch * 8) = pack8(v);
  }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pe, bf16* __restrict__ x0,
                                int B, int S, int D, int drop_on, IqRng rng, uint32_t thresh, float dscale) {
  const int nch = D / 8;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * nch) return;
  if (drop_on) rng = rng_resolve(rng);
  const int b = id / nch, ch = id % nch;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = cls[ch * 8 + e] + pe[ch * 8 + e];
  const long off = (long)b * S * D + ch * 8;
  if (drop_on) {
    const uint32_t keep = dropout_keep8(rng, (uint64_t)off >> 3, thresh);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * dscale : 0.f;
  }
  *reinterpret_cast<bf16x8*>(x0 + off) = pack8(v);
}

__global__ void embed_bwd_gather_kernel(const bf16* __restrict__ dx0, bf16* __restrict__ demb, int B, int S, int tok,
                                        int D, int cls_off, int drop_on, IqRng rng, uint32_t thresh, float dscale) {
  const int nch = D / 8;
  const long n = (long)B * tok * nch;
  if (drop_on) rng = rng_resolve(rng);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(id % nch);
    const long row = id / nch;
    const int b = (int)(row / tok), t = (int)(row % tok);
    const long off = ((long)b * S + cls_off + t) * D + ch * 8;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(dx0 + off);
    if (drop_on) {
      const uint32_t keep = dropout_keep8(rng, (uint64_t)off >> 3, thresh);
      float f[8];
      unpack8(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = ((keep >> e) & 1u) ? f[e] * dscale : 0.f;
      v = pack8(f);
    }
    *reinterpret_cast<bf16x8*>(demb + row * D + ch * 8) = v;
  }
}

// d(cls)[d] = sum_b mask(d(x0)[b,0,d]).  Block = one 8-column chunk; thread = one frame per pass (one 16 B load and one
// Philox call serve 8 columns: the old one-column-per-thread loop over frames took 36 us of pure latency), wave
// shuffles + LDS combine in fixed order.
__global__ __launch_bounds__(256) void dcls_kernel(const bf16* __restrict__ dx0, float* __restrict__ dcls, int B, int S, int D,
                                                   int drop_on, IqRng rng, uint32_t thresh, float dscale, int accumulate) {
  __shared__ float part[4][8];
  const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (drop_on) rng = rng_resolve(rng);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += 256) {
    const long off = (long)b * S * D + ch * 8;
    float v[8];
    unpack8(*reinterpret_cast<const bf16x8*>(dx0 + off), v);
    uint32_t keep = 0xffu;
    if (drop_on) keep = dropout_keep8(rng, (uint64_t)off >> 3, thresh);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += ((keep >> e) & 1u) ? (drop_on ? v[e] * dscale : v[e]) : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = wave_sum(s[e]);
    if (lane == 0) part[wave][e] = t;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    float* dst = dcls + ch * 8 + threadIdx.x;
    *dst = accumulate ? *dst + t : t;
  }
}

// ---------------------------------------------------------------------------------------------
// head: one wave per frame
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void head_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ ln_g,
                                                      const float* __restrict__ ln_b, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ feat_hat,
                                                      float* __restrict__ hstat, float* __restrict__ logits, int S,
                                                      int D, int K, int pool) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const bf16* xb = x + (long)b * S * D;
  float* fh = feat_hat + (long)b * D;
  // pooled feature -> fh (fp32)
  float s1 = 0.f;
  for (int d = lane; d < D; d += 64) {
    float f;
    if (pool == 0) {
      f = (float)xb[d];
    } else {
      f = 0.f;
      for (int s = 0; s < S; ++s) f += (float)xb[(long)s * D + d];
      f /= (float)S;
    }
    fh[d] = f;
    s1 += f;
  }
  if (ln_g) {
    const float mean = wave_sum(s1) / (float)D;
    float s2 = 0.f;
    for (int d = lane; d < D; d += 64) { const float c = fh[d] - mean; s2 += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)D + 1e-5f);
    for (int d = lane; d < D; d += 64) fh[d] = (fh[d] - mean) * rstd;
    if (lane == 0) { hstat[2 * b] = mean; hstat[2 * b + 1] = rstd; }
  }
  __syncthreads();
  if (D % 4 == 0 && D <= 1024) {
    // The K dot products, K / (kpar * 4) rounds instead of K serial ones (19 dependent memory round trips + wave
    // reductions took 23-29 us for 256 frames): the feature after the affine goes to LDS, lane group kk = lane / lpk takes
    // class k0 + kk, its lpk lanes read W[k] as float4 (coalesced), four classes per lane in flight.
    __shared__ float fs[1024];
    for (int d = lane; d < D; d += 64) fs[d] = ln_g ? ln_g[d] * fh[d] + ln_b[d] : fh[d];
    __syncthreads();
    int lpk = 64;                                   // largest power of two <= min(64, D / 4)
    while (lpk * 4 > D) lpk >>= 1;
    if (lpk >= 1) {
      const int kpar = 64 / lpk, kk = lane / lpk, dl = lane % lpk;
      for (int k0 = 0; k0 < K; k0 += kpar * 4) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = min(k0 + u * kpar + kk, K - 1);
          for (int d = 4 * dl; d < D; d += 4 * lpk) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)k * D + d);
            const f32x4 f = *reinterpret_cast<const f32x4*>(fs + d);
            a[u] += w[0] * f[0] + w[1] * f[1] + w[2] * f[2] + w[3] * f[3];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float t = a[u];
          for (int o = lpk >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
          const int k = k0 + u * kpar + kk;
          if (dl == 0 && k < K) logits[(long)b * K + k] = t + bias[k];
        }
      }
      return;
    }
  }
  for (int k = 0; k < K; ++k) {
    float a = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float f = ln_g ? ln_g[d] * fh[d] + ln_b[d] : fh[d];
      a += f * W[(long)k * D + d];
    }
    a = wave_sum(a);
    if (lane == 0) logits[(long)b * K + k] = a + bias[k];
  }
}

// label-smoothed CE; one block, thread per frame (loops if B > blockDim).  KMAX > 0: the K <= KMAX logits of a frame are
// loaded ONCE, all loads in flight together, and every pass runs on registers (three passes of dependent loads and 2 K
// expf took 13.5 us for 256 frames); KMAX = 0: any K, from memory.
template <int KMAX>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 int B, int K, float smoothing, float denom,
                                                 float* __restrict__ loss_sum, int32_t* __restrict__ n_correct,
                                                 float* __restrict__ dlogits) {
  __shared__ float sl[256];
  __shared__ int sc[256];
  float lacc = 0.f;
  int cacc = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* z = logits + (long)b * K;
    const int y = (int)labels[b];
    // a label outside [0, K) (torch raises a device-side assert): never read out of bounds; the frame's loss and
    // gradient become NaN so the mistake is loud without taking the GPU context down
    const bool y_ok = y >= 0 && y < K;
    if constexpr (KMAX > 0) {
      float zr[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) zr[k] = k < K ? z[k] : -3.0e38f;
      float mx = zr[0];
      int am = 0;
#pragma unroll
      for (int k = 1; k < KMAX; ++k)
        if (zr[k] > mx) { mx = zr[k]; am = k; }
      float se = 0.f, sz = 0.f, zy = 0.f;
      float ek[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        ek[k] = k < K ? expf(zr[k] - mx) : 0.f;
        se += ek[k];
        sz += k < K ? zr[k] : 0.f;
        zy = k == y ? zr[k] : zy;
      }
      const float lse = mx + logf(se);
      const float sum_logp = sz - (float)K * lse;
      const float nll = y_ok ? -(zy - lse) : __builtin_nanf("");
      lacc += (1.0f - smoothing) * nll + smoothing * (-sum_logp / (float)K);
      cacc += (am == y) ? 1 : 0;
      if (dlogits) {
        const float inv = 1.0f / se;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          if (k < K) {
            const float tgt = (k == y ? 1.0f - smoothing : 0.f) + smoothing / (float)K;
            dlogits[(long)b * K + k] = y_ok ? (ek[k] * inv - tgt) / denom : __builtin_nanf("");
          }
        }
      }
    } else {
      float mx = z[0];
      int am = 0;
      for (int k = 1; k < K; ++k)
        if (z[k] > mx) { mx = z[k]; am = k; }
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += expf(z[k] - mx);
      const float lse = mx + logf(se);
      float sum_logp = 0.f;
      for (int k = 0; k < K; ++k) sum_logp += z[k] - lse;
      const float nll = y_ok ? -(z[y] - lse) : __builtin_nanf("");
      lacc += (1.0f - smoothing) * nll + smoothing * (-sum_logp / (float)K);
      cacc += (am == y) ? 1 : 0;
      if (dlogits) {
        for (int k = 0; k < K; ++k) {
          const float pk = expf(z[k] - lse);
          const float tgt = (k == y ? 1.0f - smoothing : 0.f) + smoothing / (float)K;
          dlogits[(long)b * K + k] = y_ok ? (pk - tgt) / denom : __builtin_nanf("");
        }
      }
    }
  }
  sl[threadIdx.x] = lacc;
  sc[threadIdx.x] = cacc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (loss_sum) *loss_sum += sl[0];
    if (n_correct) *n_correct += sc[0];
  }
}

// d(x_L): one wave per frame; also recomputes d(feat) through the optional head LayerNorm
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat_hat,
                                                         const float* __restrict__ hstat, const float* __restrict__ ln_g,
                                                         const float* __restrict__ W, bf16* __restrict__ dx, int S,
                                                         int D, int K, int pool) {
  extern __shared__ float df[];  // [D]
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const float* dl = dlogits + (long)b * K;
  const float* fh = feat_hat + (long)b * D;
  if (threadIdx.x < 64) {       // wave 0: the feature gradient (row reductions by wave shuffles); all four waves then
    float s1 = 0.f, s2 = 0.f;   // write the frame's S x D tile (one wave doing it alone took 29 us for 19 MB)
    for (int d = lane; d < D; d += 64) {
      float a = 0.f;
      for (int k = 0; k < K; ++k) a += dl[k] * W[(long)k * D + d];
      if (ln_g) {
        a *= ln_g[d];
        s1 += a;
        s2 += a * fh[d];
      }
      df[d] = a;
    }
    if (ln_g) {
      const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
      const float rstd = hstat[2 * b + 1];
      for (int d = lane; d < D; d += 64) df[d] = rstd * (df[d] - c1 - fh[d] * c2);
    }
  }
  __syncthreads();
  bf16* dxb = dx + (long)b * S * D;
  const float inv = 1.0f / (float)S;
  for (long i = threadIdx.x; i < (long)S * (D / 8); i += 256) {
    const int s = (int)(i / (D / 8)), ch = (int)(i % (D / 8));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = pool == 0 ? (s == 0 ? df[ch * 8 + e] : 0.f) : df[ch * 8 + e] * inv;
    *reinterpret_cast<bf16x8*>(dxb + (long)s * D + ch * 8) = pack8(v);
  }
}

// dW[k,d], db[k], dln_g[d], dln_b[d]: block = 32 (k,d) outputs x 8 frame slices (+ one extra "row" k==K for
// the LN grads); the slices are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void head_bwd_w_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat_hat,
                                                         const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                         const float* __restrict__ W, float* __restrict__ dW, float* __restrict__ db,
                                                         float* __restrict__ dln_g, float* __restrict__ dln_b, int B, int D, int K,
                                                         int accumulate) {
  __shared__ float sa[8][33], sb[8][33];
  const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int id = blockIdx.x * 32 + ol;
  const int k = id / D, d = id % D;
  const bool live = k < K || (k == K && ln_g);
  float a = 0.f, b2 = 0.f;
  if (live) {
    if (k < K) {
      const float g = ln_g ? ln_g[d] : 1.f, be = ln_g ? ln_b[d] : 0.f;
#pragma unroll 4
      for (int b = sl; b < B; b += 8) {
        const float dl = dlogits[(long)b * K + k];
        a += dl * (g * feat_hat[(long)b * D + d] + be);
        b2 += dl;
      }
    } else {
      for (int b = sl; b < B; b += 8) {
        float dfn = 0.f;
        for (int kk = 0; kk < K; ++kk) dfn += dlogits[(long)b * K + kk] * W[(long)kk * D + d];
        a += dfn * feat_hat[(long)b * D + d];
        b2 += dfn;
      }
    }
  }
  sa[sl][ol] = a;
  sb[sl][ol] = b2;
  __syncthreads();
  if (sl == 0 && live) {
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ta += sa[i][ol]; tb += sb[i][ol]; }
    if (k < K) {
      dW[(long)k * D + d] = accumulate ? dW[(long)k * D + d] + ta : ta;
      if (d == 0) db[k] = accumulate ? db[k] + tb : tb;
    } else {
      dln_g[d] = accumulate ? dln_g[d] + ta : ta;
      dln_b[d] = accumulate ? dln_b[d] + tb : tb;
    }
  }
}

// The same gradients for K <= KMAX classes with the LayerNorm row folded in: block = 32 feature columns d x 8 frame
// slices, every thread keeps T[k] = sum_b dlogits[b,k] * feat_hat[b,d] and s[k] = sum_b dlogits[b,k] for ALL k, so
//   dW[k,d] = gamma[d] T[k] + beta[d] s[k],  db[k] = s[k],  dgamma[d] = sum_k W[k,d] T[k],  dbeta[d] = sum_k W[k,d] s[k]
// (the kernel above recomputed d(feat) = dlogits W per (frame, d) for the gamma / beta row: K dependent loads x B/8 frames
// per thread, 41 us for 256 frames of the raw-IQ head).  Slices are combined through LDS in fixed order.
template <int KMAX>
__global__ __launch_bounds__(256) void head_bwd_w2_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat_hat,
                                                          const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                          const float* __restrict__ W, float* __restrict__ dW, float* __restrict__ db,
                                                          float* __restrict__ dln_g, float* __restrict__ dln_b, int B, int D, int K,
                                                          int accumulate) {
  __shared__ float sT[8][KMAX][33];
  __shared__ float sS[8][KMAX];
  __shared__ float sdl[64][KMAX];                       // dlogits of 64 frames at a time, zero-padded to KMAX classes
  const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + ol;
  const bool live = d < D;
  float T[KMAX], sd[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { T[k] = 0.f; sd[k] = 0.f; }
  // (K dependent broadcast loads per frame straight from memory made this a 26 us launch for 256 frames on 6 blocks: the rows go
  //  through LDS 64 frames at a time, all their loads in flight together; the summation order per thread is unchanged)
  for (int b0 = 0; b0 < B; b0 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * KMAX; i += 256) {
      const int bb = i / KMAX, k = i - bb * KMAX;
      sdl[bb][k] = (b0 + bb < B && k < K) ? dlogits[(long)(b0 + bb) * K + k] : 0.f;
    }
    __syncthreads();
    float fv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                       // this slice's 8 frames of the group: all loads in flight together
      const int bb = sl + 8 * i;
      fv[i] = (live && b0 + bb < B) ? feat_hat[(long)(b0 + bb) * D + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int bb = sl + 8 * i;                        // (rows past B hold zeros in sdl)
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const float dl = sdl[bb][k];
        T[k] += dl * fv[i];
        sd[k] += dl;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    sT[sl][k][ol] = T[k];
    if (ol == 0) sS[sl][k] = sd[k];
  }
  __syncthreads();
  if (sl == 0 && live) {
    const float g = ln_g ? ln_g[d] : 1.f, be = ln_g ? ln_b[d] : 0.f;
    float lg = 0.f, lb = 0.f;
    for (int k = 0; k < K; ++k) {
      float t = 0.f, sk = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { t += sT[i][k][ol]; sk += sS[i][k]; }
      const float v = g * t + be * sk;
      dW[(long)k * D + d] = accumulate ? dW[(long)k * D + d] + v : v;
      if (d == 0) db[k] = accumulate ? db[k] + sk : sk;
      const float w = W[(long)k * D + d];
      lg += w * t;
      lb += w * sk;
    }
    if (ln_g) {
      dln_g[d] = accumulate ? dln_g[d] + lg : lg;
      dln_b[d] = accumulate ? dln_b[d] + lb : lb;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// optimizer
// ---------------------------------------------------------------------------------------------
constexpr int GN_BLOCKS = 1024;

__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const float* __restrict__ g, size_t n, float scale,
                                                               float* __restrict__ partial) {
  __shared__ float sm[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1] + sm[2] + sm[3]) * scale * scale;
}
__global__ __launch_bounds__(256) void gradnorm_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sm[0] + sm[1] + sm[2] + sm[3];
}

struct AdamParams {
  float lr, b1, b2, eps, wd, max_norm, grad_scale;
  int step;
  const float* gnorm_sq;
  const float* dyn;
};

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ shadow, size_t n, AdamParams a) {
  float lr = a.lr;
  int step = a.step;
  if (a.dyn) { lr = a.dyn[0]; step = (int)a.dyn[1]; }
  float coef = a.grad_scale;
  if (a.gnorm_sq && a.max_norm > 0.f) {
    const float total = sqrtf(a.gnorm_sq[0]);
    coef *= fminf(1.0f, a.max_norm / (total + 1e-6f));
  }
  const float bc1 = 1.0f - powf(a.b1, (float)step);
  const float bc2 = 1.0f - powf(a.b2, (float)step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.0f - lr * a.wd;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    bf16x4 sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * coef;
      float x = pv[e] * decay;
      mv[e] = a.b1 * mv[e] + (1.0f - a.b1) * gg;
      vv[e] = a.b2 * vv[e] + (1.0f - a.b2) * gg * gg;
      const float den = sqrtf(vv[e]) * inv_sqrt_bc2 + a.eps;
      x -= step_size * (mv[e] / den);
      pv[e] = x;
      sh[e] = (bf16)x;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (shadow) reinterpret_cast<bf16x4*>(shadow)[i] = sh;
  }
}

__global__ void counter_add_kernel(uint32_t* cu, uint32_t iu, float* cf, float incf) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (cu) *cu += iu;
    if (cf) *cf += incf;
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
    reinterpret_cast<bf16x4*>(dst)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = (bf16)src[n4 * 4 + threadIdx.x];
}

// dst[c][r] = bf16(src[r][c]); 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                             int rows, int cols) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    t[i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (c < cols && r < rows) dst[(long)c * rows + r] = (bf16)t[tx][i];
  }
}

inline int grid_for(size_t n, int per_block, int cap) {
  size_t b = (n + per_block - 1) / per_block;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

inline void fill_rng(const iq_dropout_t* d, int* on, IqRng* r, uint32_t* th, float* sc) {
  *on = 0; r->seed = 0; r->step = 0; r->site = 0; r->step_dev = nullptr; *th = 0; *sc = 1.f;
  if (d && d->p > 0.f) {
    *on = 1; r->seed = d->seed; r->step = d->step; r->site = d->site; r->step_dev = d->step_dev;
    *th = dropout_thresh(d->p); *sc = dropout_scale(d->p);
  }
}

}  // namespace

extern "C" int iq_patchify(const float* src, void* patches, int kind, int B, int C, int H, int W, int p, int Kpad,
                           iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (B <= 0) return IQ_OK;
  if (!src || !patches || C <= 0 || H <= 0 || p <= 0 || (Kpad % 8)) return IQ_ERR_ARG;
  int tok, P;
  if (kind == 0) {
    if (W <= 0) return IQ_ERR_ARG;
    tok = (H / p) * (W / p); P = C * p * p;
  } else if (kind == 1) {
    tok = H / p; P = C * p;
  } else {
    return IQ_ERR_ARG;
  }
  if (tok <= 0 || Kpad < P) return IQ_ERR_ARG;
  const size_t n = (size_t)B * tok * (Kpad / 8);
  patchify_kernel<<<grid_for(n, 256, 4096), 256, 0, (hipStream_t)stream>>>(src, (bf16*)patches, kind, B, C, H, W, p, tok,
                                                                          P, Kpad);
  return iq_launch_status();
}

extern "C" int iq_cls_rows(const float* cls, const float* pe, void* x0, int B, int S, int D, const iq_dropout_t* drop,
                           iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (B <= 0) return IQ_OK;
  if (!cls || !pe || !x0 || (D % 8)) return IQ_ERR_ARG;
  int on; IqRng r; uint32_t th; float sc;
  fill_rng(drop, &on, &r, &th, &sc);
  const int n = B * (D / 8);
  cls_rows_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(cls, pe, (bf16*)x0, B, S, D, on, r, th, sc);
  return iq_launch_status();
}

extern "C" int iq_embed_bwd_gather(const void* dx0, void* demb, float* dcls, int B, int S, int tok, int D, int has_cls,
                                   const iq_dropout_t* drop, int accumulate, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (B <= 0) return IQ_OK;
  if (!dx0 || !demb || (D % 8) || (has_cls && !dcls)) return IQ_ERR_ARG;
  int on; IqRng r; uint32_t th; float sc;
  fill_rng(drop, &on, &r, &th, &sc);
  const size_t n = (size_t)B * tok * (D / 8);
  hipStream_t st = (hipStream_t)stream;
  embed_bwd_gather_kernel<<<grid_for(n, 256, 4096), 256, 0, st>>>((const bf16*)dx0, (bf16*)demb, B, S, tok, D,
                                                                  has_cls ? 1 : 0, on, r, th, sc);
  if (has_cls) dcls_kernel<<<D / 8, 256, 0, st>>>((const bf16*)dx0, dcls, B, S, D, on, r, th, sc, accumulate);
  return iq_launch_status();
}

extern "C" int iq_head_fwd(const void* x, const float* ln_g, const float* ln_b, const float* W, const float* b,
                           float* featn, float* hstat, float* logits, int B, int S, int D, int K, int pool,
                           iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (B <= 0) return IQ_OK;
  if (!x || !W || !b || !featn || !logits || (ln_g && (!ln_b || !hstat))) return IQ_ERR_ARG;
  head_fwd_kernel<<<B, 64, 0, (hipStream_t)stream>>>((const bf16*)x, ln_g, ln_b, W, b, featn, hstat, logits, S, D, K, pool);
  return iq_launch_status();
}

extern "C" int iq_ce_fwd_bwd(const float* logits, const int64_t* labels, int B, int K, float smoothing, float denom,
                             float* loss_sum, int32_t* n_correct, float* dlogits, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (B <= 0) return IQ_OK;
  if (!logits || !labels || K <= 0 || denom <= 0.f) return IQ_ERR_ARG;
  if (K <= 32) ce_kernel<32><<<1, 256, 0, (hipStream_t)stream>>>(logits, labels, B, K, smoothing, denom, loss_sum, n_correct, dlogits);
  else ce_kernel<0><<<1, 256, 0, (hipStream_t)stream>>>(logits, labels, B, K, smoothing, denom, loss_sum, n_correct, dlogits);
  return iq_launch_status();
}

extern "C" int iq_head_bwd(const float* dlogits, const float* featn, const float* hstat, const float* ln_g,
                           const float* ln_b, const float* W, float* dW, float* db, float* dln_g, float* dln_b,
                           void* dx, int B, int S, int D, int K, int pool, int accumulate, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (B <= 0) return IQ_OK;
  if (!dlogits || !featn || !W || !dW || !db || !dx || (D % 8)) return IQ_ERR_ARG;
  if (ln_g && (!ln_b || !hstat || !dln_g || !dln_b)) return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  head_bwd_dx_kernel<<<B, 256, D * sizeof(float), st>>>(dlogits, featn, hstat, ln_g, W, (bf16*)dx, S, D, K, pool);
  if (K <= 32) {
    head_bwd_w2_kernel<32><<<(D + 31) / 32, 256, 0, st>>>(dlogits, featn, ln_g, ln_b, W, dW, db, dln_g, dln_b, B, D, K, accumulate);
  } else {
    const int n = (K + 1) * D;
    head_bwd_w_kernel<<<(n + 31) / 32, 256, 0, st>>>(dlogits, featn, ln_g, ln_b, W, dW, db, dln_g, dln_b, B, D, K, accumulate);
  }
  return iq_launch_status();
}

extern "C" size_t iq_gradnorm_ws_bytes(size_t n) { (void)n; return GN_BLOCKS * sizeof(float); }

extern "C" int iq_gradnorm_sq(const float* g, size_t n, float grad_scale, float* ws, float* out, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_OPT, stream);
  if (!g || !ws || !out || ((uintptr_t)g & 15)) return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nb = grid_for(n / 4 + 1, 256 * 4, GN_BLOCKS);
  gradnorm_partial_kernel<<<nb, 256, 0, st>>>(g, n, grad_scale, ws);
  gradnorm_final_kernel<<<1, 256, 0, st>>>(ws, nb, out);
  return iq_launch_status();
}

extern "C" int iq_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, const float* gnorm_sq,
                             float max_norm, float grad_scale, const float* dyn, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_OPT, stream);
  if (!p || !g || !m || !v || (n % 4)) return IQ_ERR_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return IQ_ERR_ARG;
  if (step < 1 && !dyn) return IQ_ERR_ARG;
  AdamParams a = {lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, step, gnorm_sq, dyn};
  adamw_kernel<<<grid_for(n / 4, 256, 2048), 256, 0, (hipStream_t)stream>>>(p, g, m, v, (bf16*)shadow_bf16, n, a);
  return iq_launch_status();
}

extern "C" int iq_counter_add(uint32_t* ctr_u32, uint32_t inc_u32, float* ctr_f32, float inc_f32, iq_stream_t stream) {
  counter_add_kernel<<<1, 64, 0, (hipStream_t)stream>>>(ctr_u32, inc_u32, ctr_f32, inc_f32);
  return iq_launch_status();
}

extern "C" int iq_cast_bf16(const float* src, void* dst, size_t n, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (n == 0) return IQ_OK;
  if (!src || !dst || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return IQ_ERR_ARG;
  cast_kernel<<<grid_for(n / 4 + 1, 256, 2048), 256, 0, (hipStream_t)stream>>>(src, (bf16*)dst, n);
  return iq_launch_status();
}

extern "C" int iq_transpose_cast_bf16(const float* src, void* dst, int rows, int cols, iq_stream_t stream) {
  IQ_PROF(IQ_FAM_MISC, stream);
  if (rows <= 0 || cols <= 0) return IQ_OK;
  if (!src || !dst) return IQ_ERR_ARG;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  transpose_cast_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, (bf16*)dst, rows, cols);
  return iq_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Input pipeline on the device: what SingleStreamImageDataset.__getitem__ does per frame on the CPU workers
// (V/dataloader/dataset.py:210-224, R/dataloader/dataset.py:214-222): per-channel z-score of the raw (len, 2) I/Q
// frame, then layout 0: [I(len) ; Q(len)] concatenated (the caller views it as (1, H, W)), layout 1: transpose to
// (2, len).  One thread per complex sample: an 8 B coalesced read, two 4 B writes at stride len.
__global__ __launch_bounds__(256) void frames_preprocess_kernel(const float* __restrict__ raw, float* __restrict__ out, long n_samp,
                                                                int len, int take, float i_mean, float i_std, float q_mean,
                                                                float q_std) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n_samp; id += (long)gridDim.x * blockDim.x) {
    const long f = id / take;
    const int t = (int)(id - f * take);
    const float2 v = *reinterpret_cast<const float2*>(raw + (f * len + t) * 2);
    out[f * 2 * take + t] = (v.x - i_mean) / i_std;           // IEEE division: bit-identical to the CPU path
    out[f * 2 * take + take + t] = (v.y - q_mean) / q_std;
  }
}

extern "C" int iq_frames_preprocess(const float* raw, float* out, int n_frames, int len, int take, const float* stats,
                                    iq_stream_t stream) {
  if (n_frames <= 0) return IQ_OK;
  if (!raw || !out || !stats || len <= 0 || take <= 0 || take > len) return IQ_ERR_ARG;
  if (!(stats[1] > 0.f) || !(stats[3] > 0.f)) return IQ_ERR_ARG;
  if ((uintptr_t)raw & 7) return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_MISC, st);
  const long n = (long)n_frames * take;
  long nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  frames_preprocess_kernel<<<(int)nb, 256, 0, st>>>(raw, out, n, len, take, stats[0], stats[1], stats[2], stats[3]);
  return iq_launch_status();
}
