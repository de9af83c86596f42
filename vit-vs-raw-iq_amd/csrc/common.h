// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes; bf16 storage, fp32 math; Philox4x32-7 counter RNG for dropout (IQ_PHILOX_ROUNDS below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define IQ_OK 0
#define IQ_ERR_ARG 1
#define IQ_ERR_UNSUPPORTED 2
#define IQ_ERR_LAUNCH 3

#define IQ_WAVE 64

static inline int iq_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? IQ_OK : IQ_ERR_LAUNCH;
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ void unpack8(const bf16x8& v, float* f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ bf16x8 pack8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
  return v;
}

// ---------------------------------------------------------------------------
// Philox4x32 (Salmon et al. 2011), 7 rounds.  counter = (group_lo, group_hi, site, step),
// key = seed.  One call yields 128 bits = 8 x 16-bit dropout decisions for the 8
// consecutive elements [8*group, 8*group+8) of a row-major activation.
// keep  <=>  u16 >= thresh, thresh = round(p * 65536)  (p quantised to 2^-16).
// Forward and backward regenerate identical bits from (seed, step, site, group).
// ---------------------------------------------------------------------------
struct IqRng {
  uint64_t seed;
  uint32_t step;
  uint32_t site;
  const uint32_t* step_dev;   // optional device-resident step (hipGraph replays bump it on device)
};

__device__ __forceinline__ IqRng rng_resolve(IqRng r) {
  if (r.step_dev) r.step = *r.step_dev;
  return r;
}

// PHILOX_ROUNDS = 7: the smallest round count the Philox authors report as passing BigCrush for
// Philox4x32 (Salmon et al., SC'11, table 2); 10 is their conservative default.  The rounds are the
// epilogues' dominant VALU cost (32-bit integer multiplies issue at quarter rate), so the 64-bit
// product form below lets the compiler use one v_mad_u64_u32 per multiplier instead of mul_lo + mul_hi.
#ifndef IQ_PHILOX_ROUNDS
#define IQ_PHILOX_ROUNDS 7
#endif
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < IQ_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)M1 * (uint64_t)c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  u32x4 o = {c0, c1, c2, c3};
  return o;
}

// 8 keep-flags (bit i = element i of the group is kept)
__device__ __forceinline__ uint32_t dropout_keep8(const IqRng& r, uint64_t group, uint32_t thresh) {
  u32x4 b = philox4x32((uint32_t)group, (uint32_t)(group >> 32), r.site, r.step,
                          (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m |= ((b[i] & 0xFFFFu) >= thresh ? 1u : 0u) << (2 * i);
    m |= ((b[i] >> 16) >= thresh ? 1u : 0u) << (2 * i + 1);
  }
  return m;
}

__host__ __device__ inline uint32_t dropout_thresh(float p) {
  float t = p * 65536.0f + 0.5f;
  if (t < 0.f) t = 0.f;
  if (t > 65535.f) t = 65535.f;
  return (uint32_t)t;
}
// scale that keeps E[x] exact for the quantised p
__host__ __device__ inline float dropout_scale(float p) {
  uint32_t t = dropout_thresh(p);
  return 65536.0f / (65536.0f - (float)t);
}

// bijective XCD-aware block remap (guide T1): consecutive logical tiles run on one XCD
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int q = nwg / NX, r = nwg % NX;
  int xcd = bid % NX, idx = bid / NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

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
