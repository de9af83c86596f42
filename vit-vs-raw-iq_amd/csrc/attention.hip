// Scaled-dot-product attention core, forward and backward, for gfx950 (MI355X).
//
// Reference: ScaleDotProductAttention.forward, V/models/layers/scale_dot_product_attention.py:23-39
//   score = q k^T / sqrt(dh); softmax(-1); score @ v            (mask never used by any caller)
// and MultiHeadAttention.split/concat, V/models/layers/multi_head_attention.py:34-47, which this
// kernel folds into its addressing: it reads the packed projection output qkv[B*S, 3D]
// (q | k | v, head h at columns h*dh) and writes out[B*S, D] already concatenated.
// Backward is the autograd backward of the same (the reference materialises and saves the S x S
// probabilities, 0.47 MB/layer/frame at S=197; here they never leave the chip: backward recomputes
// them from Q, K and the saved log-sum-exp).
//
// One workgroup (8 waves) per (frame, head).  All products are mfma_f32_16x16x32_bf16 and are
// oriented so that every accumulator tile is DIRECTLY the B operand of the product that consumes
// it (guide 3, "an accumulator tile as the next MFMA's operand"):
//   forward / backward phase B ("query on the lane"):
//        S^T[key,q] = K Q^T ;  P^T -> B operand of  O^T[d,q] = V^T P^T   (and dQ^T = K^T dS^T)
//   backward phase A ("key on the lane"):
//        S[q,key] = Q K^T ;   P, dS -> B operands of dV^T[d,key] = dO^T P and dK^T = Q^T dS
// The only transposed operands (V^T, K^T, dO^T, Q^T) come from row-major LDS images through
// ds_read_b64_tr_b16.  K-slot order inside an MFMA is free as long as A and B agree: lane group g
// uses rows {4g..4g+3} U {16+4g..16+4g+3} of each 32-row step, which is what two stacked 16x16
// accumulator tiles hold and what tr_frag() fetches.
// LDS images are [rows][dh+16] bf16 (32 B pad: conflict-free transposed reads, 2-way on row reads).
// Softmax statistics fp32, exp2 domain.  dh in {16,32,64}; dh=16 zero-pads the QK^T contraction.
#include <stdlib.h>

#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int ATT_THREADS = 512;   // 8 waves: S=197 has 7 blocks of 32 -> one pass per phase
constexpr int ATT_WAVES = ATT_THREADS / 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;

// exp2 for softmax arguments (<= 0 after the max / LSE subtraction): the bare v_exp_f32.  exp2f() wraps it in a
// denormal-range fix-up (compare, select, add, select, ldexp: five extra VALU instructions per value) that only matters
// for results below 2^-126, which are zero for every purpose here -- and the attention kernels are VALU-bound.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DH> struct AttCfg {
  static constexpr int LD = (DH == 16) ? 16 : DH + 16;  // LDS row stride (elements)
  static constexpr int KS = (DH + 31) / 32;              // 32-deep contraction steps over d
  static constexpr int DT = DH / 16;                     // 16-wide d tiles
  static constexpr int CPR = DH / 8;                     // 16 B chunks per row
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int ld, int r0, int c0, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const bf16* a = tile + (r0 + 4 * g + (i16 >> 2)) * ld + c0 + 4 * (i16 & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * ld));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// row-major fragment: 8 consecutive d of row `row`, contraction step s (d0 = 32 s + 8 g)
template <int DH>
__device__ __forceinline__ bf16x8 row_frag_lds(const bf16* tile, int row, int s, int lane) {
  const int d0 = s * 32 + 8 * (lane >> 4);
  bf16x8 v = {};
  if (DH >= 32 || d0 < DH) v = *reinterpret_cast<const bf16x8*>(tile + row * AttCfg<DH>::LD + d0);
  return v;
}
template <int DH>
__device__ __forceinline__ bf16x8 row_frag_gmem(const bf16* base, long ldg, int row, int nrows, int s, int lane) {
  const int d0 = s * 32 + 8 * (lane >> 4);
  bf16x8 v = {};
  if (row < nrows && (DH >= 32 || d0 < DH)) v = *reinterpret_cast<const bf16x8*>(base + (long)row * ldg + d0);
  return v;
}

// stage rows [r_begin, r_begin+nrows_pad) of a [S, dh] head slice (global row stride ldg) into an LDS image,
// zero-filling rows >= S
template <int DH>
__device__ __forceinline__ void stage_rows(bf16* img, const bf16* base, long ldg, int r_begin, int nrows_pad, int S,
                                           int tid) {
  constexpr int CPR = AttCfg<DH>::CPR, LD = AttCfg<DH>::LD;
  for (int id = tid; id < nrows_pad * CPR; id += ATT_THREADS) {
    const int r = id / CPR, c = id % CPR;
    bf16x8 v = {};
    if (r_begin + r < S) v = *reinterpret_cast<const bf16x8*>(base + (long)(r_begin + r) * ldg + c * 8);
    *reinterpret_cast<bf16x8*>(img + r * LD + c * 8) = v;
  }
}

// Two images at once, loads batched: every thread first REQUESTS UN 16-byte chunks of each image (2 UN loads in flight),
// then writes them.  stage_rows alone compiles to load -> s_waitcnt vmcnt(0) -> ds_write per trip (not unrolled: the trip
// count is a run-time value), i.e. 3.5 serialised memory round trips per image at S = 197: waves spend ~45 % of their
// cycles in s_waitcnt / s_barrier (SQ_WAIT_ANY, profiles/r02_pmc_attn_*.txt).  Loads are unconditional from clamped rows
// (a predicated load is merged with its zero fill at once, i.e. waited for); rows past S are zeroed when written.
template <int DH, int UN>
__device__ __forceinline__ void stage_pair(bf16* imgA, const bf16* baseA, long ldgA, bf16* imgB, const bf16* baseB, long ldgB,
                                           int r_begin, int nrows_pad, int S, int tid) {
  constexpr int CPR = AttCfg<DH>::CPR, LD = AttCfg<DH>::LD;
  const int total = nrows_pad * CPR;
  for (int base = 0; base < total; base += UN * ATT_THREADS) {
    bf16x8 va[UN], vb[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int id = min(base + j * ATT_THREADS + tid, total - 1);
      const int r = min(r_begin + id / CPR, S - 1), c = id % CPR;
      va[j] = *reinterpret_cast<const bf16x8*>(baseA + (long)r * ldgA + c * 8);
      vb[j] = *reinterpret_cast<const bf16x8*>(baseB + (long)r * ldgB + c * 8);
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int id = base + j * ATT_THREADS + tid;
      const int r = id / CPR, c = id % CPR;
      if (id < total) {
        const bool live = r_begin + r < S;
        *reinterpret_cast<bf16x8*>(imgA + r * LD + c * 8) = live ? va[j] : bf16x8{};
        *reinterpret_cast<bf16x8*>(imgB + r * LD + c * 8) = live ? vb[j] : bf16x8{};
      }
    }
  }
}

__device__ __forceinline__ bf16x8 pack_b(const f32x4& lo, const f32x4& hi) {
  bf16x8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) { v[r] = (bf16)lo[r]; v[4 + r] = (bf16)hi[r]; }
  return v;
}

__device__ __forceinline__ float group4_max(float v) {  // across lane groups (lanes l, l^16, l^32, l^48)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// MASKED: optional `mask` (uint8, 0 = masked) as (B, 1 | H, S, S): a masked position's scaled score is -10000
// (scale_dot_product_attention.py:30-31: masked_fill AFTER the 1/sqrt(dh) scaling), `masked_raw` = -10000 sqrt(dh) is
// that value in the unscaled domain the kernel keeps scores in.
template <int DH, bool MASKED>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                               float* __restrict__ lse, const uint8_t* __restrict__ mask,
                                                               long mask_hstride, int S, int H, int kchunk,
                                                               float scale_log2, float masked_raw) {
  using C = AttCfg<DH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = Ks + kchunk * C::LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int D = H * DH;
  const long ldg = 3L * D;
  const bf16* qb = qkv + (long)b * S * ldg + h * DH;
  const bf16* kb_ = qb + D;
  const bf16* vb_ = qb + 2 * D;
  const uint8_t* mk = MASKED ? mask + (long)b * (mask_hstride ? (long)H * S * S : (long)S * S) + h * mask_hstride : nullptr;
  const int qtiles = (S + 31) / 32, npass = (qtiles + ATT_WAVES - 1) / ATT_WAVES, nstage = (S + kchunk - 1) / kchunk;

  if (nstage == 1) {            // the whole sequence fits one stage: fill it first, while no accumulator is live
    stage_pair<DH, 4>(Ks, kb_, ldg, Vs, vb_, ldg, 0, min(kchunk, ((S + 31) / 32) * 32), S, tid);
    __syncthreads();
  }
  for (int pass = 0; pass < npass; ++pass) {
    const int qt = pass * ATT_WAVES + wave;
    const bool active = qt < qtiles;
    bf16x8 qf[2][C::KS];
    f32x4 o[2][C::DT];
    float m[2], lsum[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      m[u] = NEG_BIG; lsum[u] = 0.f;
#pragma unroll
      for (int s = 0; s < C::KS; ++s) qf[u][s] = row_frag_gmem<DH>(qb, ldg, qt * 32 + u * 16 + c16, active ? S : 0, s, lane);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) o[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int st = 0; st < nstage; ++st) {
      const int k0 = st * kchunk;
      const int krows = min(kchunk, ((S - k0 + 31) / 32) * 32);
      if (nstage > 1) {
        __syncthreads();
        stage_pair<DH, 2>(Ks, kb_, ldg, Vs, vb_, ldg, k0, krows, S, tid);
        __syncthreads();
      }
      if (active) {
        // wave-uniform trims (S = 197: 224 padded rows): the second 16 queries of the last tile and the second 16 keys of
        // the last block can be pure padding -- their products and softmax work are skipped; the key < S select runs only
        // in the one block that holds padding keys
        const bool u1 = qt * 32 + 16 < S;
        for (int kb = 0; kb < krows; kb += 32) {
          const bool k1 = k0 + kb + 16 < S;
          const bool kpad = k0 + kb + 32 > S;
          f32x4 sc[2][2];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1 && !k1) {
              sc[0][1] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
              continue;
            }
            bf16x8 kf[C::KS];
#pragma unroll
            for (int s = 0; s < C::KS; ++s) kf[s] = row_frag_lds<DH>(Ks, kb + kt * 16 + c16, s, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              f32x4 a = {0.f, 0.f, 0.f, 0.f};
              if (u == 0 || u1) {
#pragma unroll
                for (int s = 0; s < C::KS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[u][s], a, 0, 0, 0);
              }
              sc[u][kt] = a;
            }
          }
          bf16x8 pb[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !u1) { pb[1] = bf16x8{}; continue; }
            float mx = NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              if (kt == 1 && !k1) continue;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float x = sc[u][kt][r];               // raw score: the positive scale commutes with max and is folded
                if (MASKED) {
                  const int key = k0 + kb + kt * 16 + 4 * g + r;
                  const int q = qt * 32 + u * 16 + c16;
                  if (key < S && q < S && !mk[(long)q * S + key]) x = masked_raw;
                }
                if (kpad) {                           // into the exp2 argument below (one fma instead of mul + sub)
                  const int key = k0 + kb + kt * 16 + 4 * g + r;
                  x = key < S ? x : NEG_BIG;
                }
                sc[u][kt][r] = x;
                mx = fmaxf(mx, x);
              }
            }
            mx = group4_max(mx) * scale_log2;         // running max m[u] lives in the scaled (log2) domain
            const float mn = fmaxf(m[u], mx);
            const float alpha = fast_exp2(m[u] - mn);
            m[u] = mn;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              if (kt == 1 && !k1) continue;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float pv = fast_exp2(fmaf(sc[u][kt][r], scale_log2, -mn));
                sc[u][kt][r] = pv;
                rs += pv;
              }
            }
            lsum[u] = lsum[u] * alpha + rs;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) o[u][dt] *= alpha;
            pb[u] = pack_b(sc[u][0], sc[u][1]);
          }
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) {
            const bf16x8 vt = tr_frag(Vs, C::LD, kb, dt * 16, lane);
            o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb[0], o[0][dt], 0, 0, 0);
            if (u1) o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb[1], o[1][dt], 0, 0, 0);
          }
        }
      }
    }
    if (active) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float l = group4_sum(lsum[u]);
        const float inv = 1.0f / l;
        const int q = qt * 32 + u * 16 + c16;
        if (q < S) {
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) {
            bf16x4 w;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (bf16)(o[u][dt][r] * inv);
            *reinterpret_cast<bf16x4*>(out + ((long)b * S + q) * D + h * DH + dt * 16 + 4 * g) = w;
          }
          if (g == 0) lse[((long)b * H + h) * S + q] = (m[u] + log2f(l)) * LN2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// The staged side of each phase (phase A: Q and dO; phase B: K and V) lives in LDS `chunk` rows at a time.  One chunk
// covers the whole sequence for every training configuration (S <= 197 at dh 64, S <= 1025 at dh 16): then each image is
// staged exactly once, as before.  Longer sequences (embedding_type='conv1d', S = 1025, at dh 32 / 64: R/models/encoder.py:34-41)
// sweep the chunks once per group of 8 units / query blocks; the re-staged rows come from L2.
template <int DH, bool MASKED>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                               const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                               bf16* __restrict__ dqkv, const uint8_t* __restrict__ mask,
                                                               long mask_hstride, int S, int H, int spad, int chunk,
                                                               float scale, float masked_raw) {
  using C = AttCfg<DH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* I0 = reinterpret_cast<bf16*>(smem);            // phase A: Q   | phase B: K
  bf16* I1 = I0 + chunk * C::LD;                       // phase A: dO  | phase B: V
  float* lse_s = reinterpret_cast<float*>(I1 + chunk * C::LD);   // [spad], pre-multiplied by log2(e)
  float* del_s = lse_s + spad;                                   // [spad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int D = H * DH;
  const long ldg = 3L * D;
  const bf16* qb = qkv + (long)b * S * ldg + h * DH;
  const bf16* kb_ = qb + D;
  const bf16* vb_ = qb + 2 * D;
  const bf16* ob = out + (long)b * S * D + h * DH;
  const bf16* dob = dout + (long)b * S * D + h * DH;
  bf16* dqb = dqkv + (long)b * S * ldg + h * DH;
  const uint8_t* mk = MASKED ? mask + (long)b * (mask_hstride ? (long)H * S * S : (long)S * S) + h * mask_hstride : nullptr;
  const float scale_log2 = scale * LOG2E;
  const int nblk = spad / 32;
  const int nchunk = (spad + chunk - 1) / chunk;
  const bool single = nchunk == 1;

  // ---- delta[q] = sum_d dO*O; lse ---------------------------------------------------------------
  if (single) {
    stage_pair<DH, 4>(I0, qb, ldg, I1, dob, (long)D, 0, spad, S, tid);
    __syncthreads();
  }
  // single chunk: delta from the dO image just staged (dO is fetched from HBM once, not twice); O comes from HBM, its only use
  for (int q = tid; q < spad; q += ATT_THREADS) {
    float dl = 0.f, ls = 0.f;
    if (q < S) {
#pragma unroll
      for (int c = 0; c < C::CPR; ++c) {
        const bf16x8 a = single ? *reinterpret_cast<const bf16x8*>(I1 + q * C::LD + c * 8)
                                : *reinterpret_cast<const bf16x8*>(dob + (long)q * D + c * 8);
        const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(ob + (long)q * D + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)a[e] * (float)o8[e];
      }
      ls = lse[((long)b * H + h) * S + q] * LOG2E;
    }
    del_s[q] = dl;
    lse_s[q] = ls;
  }
  __syncthreads();

  // ---- phase A: wave owns 16 keys, sweeps queries; dV^T, dK^T in registers ---------------------
  // (16-key units keep the kernel at <= 128 VGPRs, so two 8-wave workgroups share a CU and one's staging /
  //  dependency stalls overlap the other's MFMAs; 32-key units needed 195 VGPRs = one workgroup per CU.)
  const int nunit = (S + 15) / 16;      // units made only of padding keys are skipped
  for (int ug = 0; ug * ATT_WAVES < nunit; ++ug) {
    const int unit = ug * ATT_WAVES + wave;
    const bool have = unit < nunit;       // wave-uniform
    bf16x8 kf[C::KS], vf[C::KS];
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      kf[s] = row_frag_gmem<DH>(kb_, ldg, unit * 16 + c16, have ? S : 0, s, lane);
      vf[s] = row_frag_gmem<DH>(vb_, ldg, unit * 16 + c16, have ? S : 0, s, lane);
    }
    f32x4 dv[C::DT], dk[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int key = unit * 16 + c16;
    const bool unit_partial = unit * 16 + 16 > S;       // wave-uniform
    for (int qc = 0; qc < nchunk; ++qc) {
      const int q0 = qc * chunk;
      const int rows = min(chunk, spad - q0);
      if (!single) {
        __syncthreads();
        stage_pair<DH, 2>(I0, qb, ldg, I1, dob, (long)D, q0, rows, S, tid);
        __syncthreads();
      }
      if (!have) continue;
      for (int ql = 0; ql < rows; ql += 32) {
        const bool u1 = q0 + ql + 16 < S;      // wave-uniform: the block's second 16 queries are not all padding
        f32x4 p[2], ds[2];   // [u]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !u1) { p[1] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < C::KS; ++s) {
            const bf16x8 qa = row_frag_lds<DH>(I0, ql + u * 16 + c16, s, lane);
            const bf16x8 da = row_frag_lds<DH>(I1, ql + u * 16 + c16, s, lane);
            sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[s], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[s], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = q0 + ql + u * 16 + 4 * g + r;
            // Padded QUERY rows need no mask: Q = dO = 0 and lse = delta = 0 there, so P = 1 and dS = 0 multiply zeros.
            // Padded KEYS do (only this wave's unit can hold them): P = exp2(-lse) is unbounded when a row's scores are
            // all very negative, and inf * 0 would poison dQ.
            bool blocked = false;
            if (MASKED) blocked = q < S && key < S && !mk[(long)q * S + key];
            float pv = fast_exp2((blocked ? masked_raw : sa[r]) * scale_log2 - lse_s[q]);
            if (unit_partial) pv = key < S ? pv : 0.f;
            p[u][r] = pv;
            ds[u][r] = blocked ? 0.f : pv * (dp[r] - del_s[q]) * scale;     // masked_fill passes no gradient
          }
        }
        const bf16x8 pB = pack_b(p[0], p[1]), dsB = pack_b(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const bf16x8 doT = tr_frag(I1, C::LD, ql, dt * 16, lane);
          const bf16x8 qT = tr_frag(I0, C::LD, ql, dt * 16, lane);
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT, pB, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, dsB, dk[dt], 0, 0, 0);
        }
      }
    }
    if (have && key < S) {
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        bf16x4 wk, wv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { wk[r] = (bf16)dk[dt][r]; wv[r] = (bf16)dv[dt][r]; }
        *reinterpret_cast<bf16x4*>(dqb + (long)key * ldg + D + dt * 16 + 4 * g) = wk;
        *reinterpret_cast<bf16x4*>(dqb + (long)key * ldg + 2 * D + dt * 16 + 4 * g) = wv;
      }
    }
  }

  // ---- phase B: restage K, V; wave owns 32 queries, sweeps keys; dQ^T in registers ---------------
  // Single chunk: the wave's query-side fragments (its first query block) come out of the Q / dO images before K and V
  // overwrite them: Q and dO are fetched from HBM once.  (Later blocks of a wave, and every block of a chunked run, use
  // global loads.)
  bf16x8 qf[2][C::KS], dof[2][C::KS];
  if (single) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < C::KS; ++s) {
        const int q = min(wave, nblk - 1) * 32 + u * 16 + c16;
        qf[u][s] = row_frag_lds<DH>(I0, q, s, lane);
        dof[u][s] = row_frag_lds<DH>(I1, q, s, lane);
      }
  }
  for (int qg = 0; qg * ATT_WAVES < nblk; ++qg) {
    const int qblk = qg * ATT_WAVES + wave;
    const bool have = qblk < nblk;          // wave-uniform
    float lq[2], dl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = min(qblk, nblk - 1) * 32 + u * 16 + c16;
      if (!single || qg != 0) {
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
          qf[u][s] = row_frag_gmem<DH>(qb, ldg, q, have ? S : 0, s, lane);
          dof[u][s] = row_frag_gmem<DH>(dob, (long)D, q, have ? S : 0, s, lane);
        }
      }
      lq[u] = lse_s[q];
      dl[u] = del_s[q];
    }
    f32x4 dq[C::DT][2];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u) dq[dt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nchunk; ++kc) {
      const int k0 = kc * chunk;
      const int rows = min(chunk, spad - k0);
      if (!single || qg == 0) {
        __syncthreads();
        stage_pair<DH, 2>(I0, kb_, ldg, I1, vb_, ldg, k0, rows, S, tid);
        __syncthreads();
      }
      if (!have) continue;
      const bool u1 = qblk * 32 + 16 < S;               // wave-uniform: this block's second 16 queries are not all padding
      for (int kl = 0; kl < rows; kl += 32) {
        const bool kblk_partial = k0 + kl + 32 > S;     // wave-uniform
        const bool k1 = k0 + kl + 16 < S;               // the block's second 16 keys are not all padding
        f32x4 ds[2][2];  // [u][kt]
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          if (kt == 1 && !k1) { ds[0][1] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[1][1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
          bf16x8 ka[C::KS], va[C::KS];
#pragma unroll
          for (int s = 0; s < C::KS; ++s) {
            ka[s] = row_frag_lds<DH>(I0, kl + kt * 16 + c16, s, lane);
            va[s] = row_frag_lds<DH>(I1, kl + kt * 16 + c16, s, lane);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !u1) { ds[1][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
              sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[s], qf[u][s], sa, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[s], dof[u][s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = k0 + kl + kt * 16 + 4 * g + r;
              bool blocked = false;
              if (MASKED) {
                const int q = qblk * 32 + u * 16 + c16;
                blocked = q < S && key < S && !mk[(long)q * S + key];
              }
              float pv = fast_exp2((blocked ? masked_raw : sa[r]) * scale_log2 - lq[u]);
              if (kblk_partial) pv = key < S ? pv : 0.f;          // padded keys: see phase A
              ds[u][kt][r] = blocked ? 0.f : pv * (dp[r] - dl[u]) * scale;
            }
          }
        }
        bf16x8 dsB[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) dsB[u] = pack_b(ds[u][0], ds[u][1]);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const bf16x8 kT = tr_frag(I0, C::LD, kl, dt * 16, lane);
          dq[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsB[0], dq[dt][0], 0, 0, 0);
          if (u1) dq[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsB[1], dq[dt][1], 0, 0, 0);
        }
      }
    }
    if (have) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = qblk * 32 + u * 16 + c16;
        if (q < S) {
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) {
            bf16x4 w;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (bf16)dq[dt][u][r];
            *reinterpret_cast<bf16x4*>(dqb + (long)q * ldg + dt * 16 + 4 * g) = w;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Per-FRAME kernels for short sequences (cfg C: S = 65, 8 heads of 16)
// ---------------------------------------------------------------------------------------------
// One workgroup per (frame, head) is the wrong shape when S is a few 32-row tiles: at S = 65 the forward has 3 query tiles
// for 8 waves, the backward 5 key units / 3 query blocks, every workgroup stages 65 x 32-byte row pieces of q, k and v out
// of 768-byte rows, and 2,048 such workgroups cost a launch round each (0.8-0.9 TB/s, 0.10 of peak in round 2).
// Here ONE workgroup owns a frame: its packed qkv rows [S][3D] are one contiguous block of global memory, staged once
// with fully coalesced 16-byte loads into an LDS image [spad][3D + 16] (row stride = 32 B mod 256 B: the row-fragment
// ds_read_b128 and the transposing ds_read_b64_tr_b16 are both conflict-free), and wave w then runs heads w, w + 8, ...
// entirely on its own: same products, same orientation, same softmax arithmetic as the kernels above, no barrier after
// the staging one.  Query / key halves that are pure padding (S = 65: rows 80..95) are skipped.
template <int DH>
__device__ __forceinline__ bf16x8 row_frag_img(const bf16* tile, int ld, int row, int s, int lane) {
  const int d0 = s * 32 + 8 * (lane >> 4);
  bf16x8 v = {};
  if (DH >= 32 || d0 < DH) v = *reinterpret_cast<const bf16x8*>(tile + row * ld + d0);
  return v;
}

// rows [0, spad) x `cols` elements (cols % 8 == 0) of a contiguous [S][cols] global block -> LDS image with row stride ld;
// rows >= S are zero-filled.  UN loads in flight per thread.
template <int UN>
__device__ __forceinline__ void stage_block(bf16* img, int ld, const bf16* src, int cols, int S, int spad, int tid, int nthr) {
  const int cpr = cols >> 3, total = spad * cpr, live = S * cpr;
  for (int base = 0; base < total; base += UN * nthr) {
    bf16x8 v[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int id = min(base + j * nthr + tid, live - 1);
      v[j] = *reinterpret_cast<const bf16x8*>(src + (long)id * 8);
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int id = base + j * nthr + tid;
      if (id < total) {
        const int r = id / cpr, c = id - r * cpr;
        *reinterpret_cast<bf16x8*>(img + r * ld + c * 8) = id < live ? v[j] : bf16x8{};
      }
    }
  }
}

// G = query tiles of a head that one wave carries through the key sweep TOGETHER: their softmax chains (MFMA -> max ->
// shuffles -> exp2 -> pack -> MFMA) are independent, so the compiler interleaves them -- with one tile at a time and two
// waves per SIMD the sweep was a serial dependency chain (15 us for cfg C's 256 frames) -- and a key block's K / V
// fragments are read from LDS once for all of them.
template <int DH, int G>
__global__ __launch_bounds__(ATT_THREADS) void attn_frame_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                     float* __restrict__ lse, int S, int H, int spad,
                                                                     float scale_log2) {
  using C = AttCfg<DH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* img = reinterpret_cast<bf16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int nwave = blockDim.x >> 6;
  const int b = blockIdx.x;
  const int D = H * DH, ld = 3 * D + 16;
  stage_block<8>(img, ld, qkv + (long)b * S * 3 * D, 3 * D, S, spad, tid, blockDim.x);
  __syncthreads();
  const int qtiles = (S + 31) / 32;
  for (int h = wave; h < H; h += nwave) {
    const bf16* Qs = img + h * DH;
    const bf16* Ks = Qs + D;
    const bf16* Vs = Qs + 2 * D;
    for (int qt0 = 0; qt0 < qtiles; qt0 += G) {
      bf16x8 qf[G][2][C::KS];
      f32x4 o[G][2][C::DT];
      float m[G][2], lsum[G][2];
      bool live[G][2];                              // wave-uniform: this 16-query half holds at least one real query
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q0 = (qt0 + t) * 32 + u * 16;
          live[t][u] = q0 < S;
          m[t][u] = NEG_BIG; lsum[t][u] = 0.f;
#pragma unroll
          for (int s = 0; s < C::KS; ++s) qf[t][u][s] = row_frag_img<DH>(Qs, ld, min(q0 + c16, spad - 1), s, lane);
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) o[t][u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      for (int kb = 0; kb < spad; kb += 32) {
        const bool k1 = kb + 16 < S;                // wave-uniform: the second 16 keys of the block are not all padding
        bf16x8 kf[2][C::KS], vt[C::DT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s = 0; s < C::KS; ++s) kf[kt][s] = row_frag_img<DH>(Ks, ld, kb + kt * 16 + c16, s, lane);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) vt[dt] = tr_frag(Vs, ld, kb, dt * 16, lane);
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (!live[t][u]) continue;
            f32x4 sc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              f32x4 a = {0.f, 0.f, 0.f, 0.f};
              if (kt == 0 || k1) {
#pragma unroll
                for (int s = 0; s < C::KS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][s], qf[t][u][s], a, 0, 0, 0);
              }
              sc[kt] = a;
            }
            float mx = NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              if (kt == 1 && !k1) { sc[1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int key = kb + kt * 16 + 4 * g + r;
                const float x = key < S ? sc[kt][r] : NEG_BIG;
                sc[kt][r] = x;
                mx = fmaxf(mx, x);
              }
            }
            mx = group4_max(mx) * scale_log2;
            const float mn = fmaxf(m[t][u], mx);
            const float alpha = fast_exp2(m[t][u] - mn);
            m[t][u] = mn;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              if (kt == 1 && !k1) continue;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float pv = fast_exp2(fmaf(sc[kt][r], scale_log2, -mn));
                sc[kt][r] = pv;
                rs += pv;
              }
            }
            lsum[t][u] = lsum[t][u] * alpha + rs;
            const bf16x8 pb = pack_b(sc[0], sc[1]);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
              o[t][u][dt] *= alpha;
              o[t][u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[dt], pb, o[t][u][dt], 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (!live[t][u]) continue;
          const float l = group4_sum(lsum[t][u]);
          const float inv = 1.0f / l;
          const int q = (qt0 + t) * 32 + u * 16 + c16;
          if (q < S) {
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
              bf16x4 w;
#pragma unroll
              for (int r = 0; r < 4; ++r) w[r] = (bf16)(o[t][u][dt][r] * inv);
              *reinterpret_cast<bf16x4*>(out + ((long)b * S + q) * D + h * DH + dt * 16 + 4 * g) = w;
            }
            if (g == 0) lse[((long)b * H + h) * S + q] = (m[t][u] + log2f(l)) * LN2;
          }
        }
    }
  }
}

// Backward of the same: qkv image + dO image [spad][D + 16] + per-head lse / delta rows, everything staged once; wave w
// runs phase A (16-key units: dK, dV) and phase B (32-query blocks: dQ) of heads w, w + 8, ... from the images, UG key
// units / QG query blocks at a time (independent chains for the scheduler, shared fragment reads), as the forward does.
template <int DH, int UG, int QG>
__global__ __launch_bounds__(ATT_THREADS) void attn_frame_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                                     const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                     bf16* __restrict__ dqkv, int S, int H, int spad, float scale) {
  using C = AttCfg<DH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int nwave = blockDim.x >> 6;
  const int b = blockIdx.x;
  const int D = H * DH, ld = 3 * D + 16, ldo = D + 16;
  bf16* img = reinterpret_cast<bf16*>(smem);
  bf16* dimg = img + spad * ld;
  float* lse_s = reinterpret_cast<float*>(dimg + spad * ldo);    // [H][spad], pre-multiplied by log2(e)
  float* del_s = lse_s + H * spad;                                // [H][spad]
  const long ldg = 3L * D;
  const float scale_log2 = scale * LOG2E;
  // delta[h][q] = sum_d dO * O over the head's columns (O: its only use, straight from global memory; consecutive threads
  // take consecutive heads of one row: coalesced 2 * DH-byte pieces) and lse -- requested first, consumed after the images
  // are written, so that all of the kernel's input is in flight at once
  for (int id = tid; id < spad * H; id += blockDim.x) {
    const int q = id / H, h = id - q * H;
    float dl = 0.f, ls = 0.f;
    if (q < S) {
      const bf16* orow = out + ((long)b * S + q) * D + h * DH;
      const bf16* drow = dout + ((long)b * S + q) * D + h * DH;
#pragma unroll
      for (int c = 0; c < C::CPR; ++c) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(drow + c * 8);
        const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(orow + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)a[e] * (float)o8[e];
      }
      ls = lse[((long)b * H + h) * S + q] * LOG2E;
    }
    del_s[h * spad + q] = dl;
    lse_s[h * spad + q] = ls;
  }
  stage_block<8>(img, ld, qkv + (long)b * S * ldg, 3 * D, S, spad, tid, blockDim.x);
  stage_block<4>(dimg, ldo, dout + (long)b * S * D, D, S, spad, tid, blockDim.x);
  __syncthreads();
  bf16* dqb = dqkv + (long)b * S * ldg;
  const int nunit = (S + 15) / 16, nblk = spad / 32;
  for (int h = wave; h < H; h += nwave) {
    const bf16* Qs = img + h * DH;
    const bf16* Ks = Qs + D;
    const bf16* Vs = Qs + 2 * D;
    const bf16* Os = dimg + h * DH;
    const float* lq_s = lse_s + h * spad;
    const float* dl_s = del_s + h * spad;
    // ---- phase A: 16 keys on the lanes, sweep the queries: dV^T, dK^T ------------------------------------------------
    for (int un0 = 0; un0 < nunit; un0 += UG) {
      bf16x8 kf[UG][C::KS], vf[UG][C::KS];
      f32x4 dv[UG][C::DT], dk[UG][C::DT];
#pragma unroll
      for (int j = 0; j < UG; ++j) {
        const int row = min((un0 + j) * 16 + c16, spad - 1);
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
          kf[j][s] = row_frag_img<DH>(Ks, ld, row, s, lane);
          vf[j][s] = row_frag_img<DH>(Vs, ld, row, s, lane);
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) { dv[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      for (int ql = 0; ql < spad; ql += 32) {
        const bool u1 = ql + 16 < S;                      // wave-uniform: second 16 queries not all padding
        bf16x8 qa[2][C::KS], da[2][C::KS], doT[C::DT], qT[C::DT];
        f32x4 lq4[2], dl4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int s = 0; s < C::KS; ++s) {
            qa[u][s] = row_frag_img<DH>(Qs, ld, ql + u * 16 + c16, s, lane);
            da[u][s] = row_frag_img<DH>(Os, ldo, ql + u * 16 + c16, s, lane);
          }
          lq4[u] = *reinterpret_cast<const f32x4*>(lq_s + ql + u * 16 + 4 * g);
          dl4[u] = *reinterpret_cast<const f32x4*>(dl_s + ql + u * 16 + 4 * g);
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          doT[dt] = tr_frag(Os, ldo, ql, dt * 16, lane);
          qT[dt] = tr_frag(Qs, ld, ql, dt * 16, lane);
        }
#pragma unroll
        for (int j = 0; j < UG; ++j) {
          const int unit = un0 + j;
          if (unit >= nunit) continue;                    // wave-uniform
          const int key = unit * 16 + c16;
          const bool unit_partial = unit * 16 + 16 > S;   // wave-uniform
          f32x4 p[2], ds[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !u1) { p[1] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
              sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[u][s], kf[j][s], sa, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[u][s], vf[j][s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // padded QUERY rows: Q = dO = 0 and lse = delta = 0, so P = 1 and dS = 0 multiply zeros; padded KEYS are masked
              float pv = fast_exp2(sa[r] * scale_log2 - lq4[u][r]);
              if (unit_partial) pv = key < S ? pv : 0.f;
              p[u][r] = pv;
              ds[u][r] = pv * (dp[r] - dl4[u][r]) * scale;
            }
          }
          const bf16x8 pB = pack_b(p[0], p[1]), dsB = pack_b(ds[0], ds[1]);
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) {
            dv[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT[dt], pB, dv[j][dt], 0, 0, 0);
            dk[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[dt], dsB, dk[j][dt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < UG; ++j) {
        const int key = (un0 + j) * 16 + c16;
        if (un0 + j < nunit && key < S) {
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) {
            bf16x4 wk, wv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { wk[r] = (bf16)dk[j][dt][r]; wv[r] = (bf16)dv[j][dt][r]; }
            *reinterpret_cast<bf16x4*>(dqb + (long)key * ldg + D + h * DH + dt * 16 + 4 * g) = wk;
            *reinterpret_cast<bf16x4*>(dqb + (long)key * ldg + 2 * D + h * DH + dt * 16 + 4 * g) = wv;
          }
        }
      }
    }
    // ---- phase B: 32 queries on the lanes, sweep the keys: dQ^T ------------------------------------------------------
    for (int qb0 = 0; qb0 < nblk; qb0 += QG) {
      bf16x8 qf[QG][2][C::KS], dof[QG][2][C::KS];
      float lq[QG][2], dl[QG][2];
      bool live[QG][2];
      f32x4 dq[QG][C::DT][2];
#pragma unroll
      for (int t = 0; t < QG; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q0 = (qb0 + t) * 32 + u * 16;
          live[t][u] = q0 < S;
          const int q = min(q0 + c16, spad - 1);
#pragma unroll
          for (int s = 0; s < C::KS; ++s) {
            qf[t][u][s] = row_frag_img<DH>(Qs, ld, q, s, lane);
            dof[t][u][s] = row_frag_img<DH>(Os, ldo, q, s, lane);
          }
          lq[t][u] = lq_s[q];
          dl[t][u] = dl_s[q];
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) dq[t][dt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      for (int kl = 0; kl < spad; kl += 32) {
        const bool kblk_partial = kl + 32 > S;            // wave-uniform
        const bool k1 = kl + 16 < S;
        bf16x8 ka[2][C::KS], va[2][C::KS], kT[C::DT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s = 0; s < C::KS; ++s) {
            ka[kt][s] = row_frag_img<DH>(Ks, ld, kl + kt * 16 + c16, s, lane);
            va[kt][s] = row_frag_img<DH>(Vs, ld, kl + kt * 16 + c16, s, lane);
          }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) kT[dt] = tr_frag(Ks, ld, kl, dt * 16, lane);
#pragma unroll
        for (int t = 0; t < QG; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (!live[t][u]) continue;
            f32x4 ds[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              if (kt == 1 && !k1) { ds[1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
              f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int s = 0; s < C::KS; ++s) {
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kt][s], qf[t][u][s], sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[kt][s], dof[t][u][s], dp, 0, 0, 0);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int key = kl + kt * 16 + 4 * g + r;
                float pv = fast_exp2(sa[r] * scale_log2 - lq[t][u]);
                if (kblk_partial) pv = key < S ? pv : 0.f;
                ds[kt][r] = pv * (dp[r] - dl[t][u]) * scale;
              }
            }
            const bf16x8 dsB = pack_b(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
              dq[t][dt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[dt], dsB, dq[t][dt][u], 0, 0, 0);
          }
      }
#pragma unroll
      for (int t = 0; t < QG; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = (qb0 + t) * 32 + u * 16 + c16;
          if (live[t][u] && q < S) {
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
              bf16x4 w;
#pragma unroll
              for (int r = 0; r < 4; ++r) w[r] = (bf16)dq[t][dt][u][r];
              *reinterpret_cast<bf16x4*>(dqb + (long)q * ldg + h * DH + dt * 16 + 4 * g) = w;
            }
          }
        }
    }
  }
}

// LDS bytes of the per-frame kernels; 0 = not their shape (sequence too long for one image)
constexpr size_t ATT_FRAME_LDS_MAX = 112 * 1024;   // larger frames measured no faster than (frame, head) workgroups (profiles/r03_probes.txt)
inline size_t frame_fwd_lds(int S, int H, int dh) {
  const int spad = (S + 31) / 32 * 32;
  return (size_t)spad * (3 * H * dh + 16) * 2;
}
inline size_t frame_bwd_lds(int S, int H, int dh) {
  const int spad = (S + 31) / 32 * 32;
  return (size_t)spad * (3 * H * dh + 16) * 2 + (size_t)spad * (H * dh + 16) * 2 + (size_t)2 * H * spad * sizeof(float);
}
// Per-frame when the (frame, head) kernels would idle most of their 8 waves: at most 4 query tiles.  IQ_TUNE_ATTN_FRAME=0|1
// forces the choice where both apply (probes).
inline bool use_frame(int S, size_t lds) {
  static const int tune = [] { const char* e = getenv("IQ_TUNE_ATTN_FRAME"); return e ? atoi(e) : -1; }();
  if (lds > ATT_FRAME_LDS_MAX) return false;
  if (tune == 0) return false;
  if (tune == 1) return true;
  return S <= 128;
}

template <int DH>
int launch_frame_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, hipStream_t st) {
  const int spad = (S + 31) / 32 * 32;
  const size_t lds = frame_fwd_lds(S, H, DH);
  auto k = attn_frame_fwd_kernel<DH, (DH == 64 ? 2 : 4)>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = 64 * (H < ATT_WAVES ? H : ATT_WAVES);
  k<<<B, threads, lds, st>>>((const bf16*)qkv, (bf16*)out, lse, S, H, spad, LOG2E / sqrtf((float)DH));
  return iq_launch_status();
}
template <int DH>
int launch_frame_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S, int H,
                     hipStream_t st) {
  const int spad = (S + 31) / 32 * 32;
  const size_t lds = frame_bwd_lds(S, H, DH);
  auto k = attn_frame_bwd_kernel<DH, (DH == 64 ? 2 : 3), (DH == 16 ? 4 : DH == 32 ? 3 : 2)>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = 64 * (H < ATT_WAVES ? H : ATT_WAVES);
  k<<<B, threads, lds, st>>>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, S, H, spad,
                             1.0f / sqrtf((float)DH));
  return iq_launch_status();
}

constexpr size_t ATT_LDS_FWD_BUDGET = 72 * 1024;   // S=197, dh=64 (224 rows) in one stage, 2 WG/CU
constexpr size_t ATT_LDS_BWD_BUDGET = 76 * 1024;   // S=197, dh=64: both 224-row images + statistics, 2 WG/CU
constexpr int ATT_MAX_S = 4096;                    // lse / delta stay whole in LDS (8 B per padded row)

// rows of the staged side kept in LDS at a time (multiple of 32)
template <int DH> int bwd_chunk_rows(int S) {
  const int spad = (S + 31) / 32 * 32;
  const long room = (long)ATT_LDS_BWD_BUDGET - (long)2 * spad * sizeof(float);
  int chunk = (int)(room / (2 * AttCfg<DH>::LD * 2)) / 32 * 32;
  if (chunk > spad) chunk = spad;
  return chunk;
}
template <int DH> size_t bwd_lds_bytes(int S) {
  const int spad = (S + 31) / 32 * 32;
  return (size_t)2 * bwd_chunk_rows<DH>(S) * AttCfg<DH>::LD * 2 + (size_t)2 * spad * sizeof(float);
}

template <int DH, bool MASKED>
int launch_fwd(const void* qkv, void* out, float* lse, const uint8_t* mask, long mask_hstride, int B, int S, int H,
               hipStream_t st) {
  const int spad = (S + 31) / 32 * 32;
  int kchunk = (int)(ATT_LDS_FWD_BUDGET / (2 * AttCfg<DH>::LD * 2)) / 32 * 32;
  if (kchunk > spad) kchunk = spad;
  const size_t lds = (size_t)2 * kchunk * AttCfg<DH>::LD * 2;
  const float scale_log2 = LOG2E / sqrtf((float)DH);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<DH, MASKED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_fwd_kernel<DH, MASKED><<<B * H, ATT_THREADS, lds, st>>>((const bf16*)qkv, (bf16*)out, lse, mask, mask_hstride, S, H,
                                                               kchunk, scale_log2, -10000.0f * sqrtf((float)DH));
  return iq_launch_status();
}

template <int DH, bool MASKED>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const uint8_t* mask,
               long mask_hstride, int B, int S, int H, hipStream_t st) {
  const int spad = (S + 31) / 32 * 32;
  const int chunk = bwd_chunk_rows<DH>(S);
  if (S > ATT_MAX_S || chunk < 32) return IQ_ERR_UNSUPPORTED;
  const size_t lds = bwd_lds_bytes<DH>(S);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<DH, MASKED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_kernel<DH, MASKED><<<B * H, ATT_THREADS, lds, st>>>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse,
                                                               (bf16*)dqkv, mask, mask_hstride, S, H, spad, chunk,
                                                               1.0f / sqrtf((float)DH), -10000.0f * sqrtf((float)DH));
  return iq_launch_status();
}

template <bool MASKED>
int dispatch_fwd(const void* qkv, void* out, float* lse, const uint8_t* mask, long hs, int B, int S, int H, int dh,
                 hipStream_t st) {
  switch (dh) {
    case 16: return launch_fwd<16, MASKED>(qkv, out, lse, mask, hs, B, S, H, st);
    case 32: return launch_fwd<32, MASKED>(qkv, out, lse, mask, hs, B, S, H, st);
    case 64: return launch_fwd<64, MASKED>(qkv, out, lse, mask, hs, B, S, H, st);
    default: return IQ_ERR_UNSUPPORTED;
  }
}
template <bool MASKED>
int dispatch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const uint8_t* mask,
                 long hs, int B, int S, int H, int dh, hipStream_t st) {
  switch (dh) {
    case 16: return launch_bwd<16, MASKED>(qkv, out, dout, lse, dqkv, mask, hs, B, S, H, st);
    case 32: return launch_bwd<32, MASKED>(qkv, out, dout, lse, dqkv, mask, hs, B, S, H, st);
    case 64: return launch_bwd<64, MASKED>(qkv, out, dout, lse, dqkv, mask, hs, B, S, H, st);
    default: return IQ_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int iq_attn_supported(int S, int dh) {
  if (S <= 0 || S > ATT_MAX_S) return 0;
  return (dh == 16 || dh == 32 || dh == 64) ? 1 : 0;
}

extern "C" int iq_attn_fwd_masked(const void* qkv, void* out, float* lse, const uint8_t* mask, long mask_hstride, int B,
                                  int S, int H, int dh, iq_stream_t stream) {
  if (B <= 0) return IQ_OK;
  if (!qkv || !out || !lse || S <= 0 || H <= 0) return IQ_ERR_ARG;
  if (mask_hstride != 0 && mask_hstride != (long)S * S) return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_ATTN_FWD, st);
  const bool frame = !mask && (dh == 16 || dh == 32 || dh == 64) && use_frame(S, frame_fwd_lds(S, H, dh));
  {
    const double rows = (double)B * S, Dm = (double)H * dh;
    if (frame) IQ_PROF_K(2.0 * rows * 4.0 * Dm + 4.0 * B * H * S, 4.0 * B * H * (double)S * S * dh, "attn_frame_fwd_kernel<%d>", dh);
    else IQ_PROF_K(2.0 * rows * 4.0 * Dm + 4.0 * B * H * S, 4.0 * B * H * (double)S * S * dh, "attn_fwd_kernel<%d, %s>", dh, mask ? "true" : "false");
  }
  if (frame) {
    switch (dh) {
      case 16: return launch_frame_fwd<16>(qkv, out, lse, B, S, H, st);
      case 32: return launch_frame_fwd<32>(qkv, out, lse, B, S, H, st);
      default: return launch_frame_fwd<64>(qkv, out, lse, B, S, H, st);
    }
  }
  return mask ? dispatch_fwd<true>(qkv, out, lse, mask, mask_hstride, B, S, H, dh, st)
              : dispatch_fwd<false>(qkv, out, lse, nullptr, 0, B, S, H, dh, st);
}
extern "C" int iq_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int dh, iq_stream_t stream) {
  return iq_attn_fwd_masked(qkv, out, lse, nullptr, 0, B, S, H, dh, stream);
}

extern "C" int iq_attn_bwd_masked(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                  const uint8_t* mask, long mask_hstride, int B, int S, int H, int dh,
                                  iq_stream_t stream) {
  if (B <= 0) return IQ_OK;
  if (!qkv || !out || !dout || !lse || !dqkv || S <= 0 || H <= 0) return IQ_ERR_ARG;
  if (mask_hstride != 0 && mask_hstride != (long)S * S) return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_ATTN_BWD, st);
  const bool frame = !mask && (dh == 16 || dh == 32 || dh == 64) && use_frame(S, frame_bwd_lds(S, H, dh));
  {
    const double rows = (double)B * S, Dm = (double)H * dh;      // qkv, out, dout read; dqkv written; 7 MFMA products
    if (frame) IQ_PROF_K(2.0 * rows * 8.0 * Dm + 4.0 * B * H * S, 14.0 * B * H * (double)S * S * dh, "attn_frame_bwd_kernel<%d>", dh);
    else IQ_PROF_K(2.0 * rows * 8.0 * Dm + 4.0 * B * H * S, 14.0 * B * H * (double)S * S * dh, "attn_bwd_kernel<%d, %s>", dh, mask ? "true" : "false");
  }
  if (frame) {
    switch (dh) {
      case 16: return launch_frame_bwd<16>(qkv, out, dout, lse, dqkv, B, S, H, st);
      case 32: return launch_frame_bwd<32>(qkv, out, dout, lse, dqkv, B, S, H, st);
      default: return launch_frame_bwd<64>(qkv, out, dout, lse, dqkv, B, S, H, st);
    }
  }
  return mask ? dispatch_bwd<true>(qkv, out, dout, lse, dqkv, mask, mask_hstride, B, S, H, dh, st)
              : dispatch_bwd<false>(qkv, out, dout, lse, dqkv, nullptr, 0, B, S, H, dh, st);
}
extern "C" int iq_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                           int S, int H, int dh, iq_stream_t stream) {
  return iq_attn_bwd_masked(qkv, out, dout, lse, dqkv, nullptr, 0, B, S, H, dh, stream);
}
