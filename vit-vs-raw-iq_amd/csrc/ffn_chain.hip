// The encoder layer from the attention output on as ONE launch per direction (gfx950).  Core: the position-wise feed-forward
// sub-layer + norm2, the hidden activation never re-read:
//
//     H[M,F]  = dropout1(relu(X1[M,D] * W1[F,D]^T + b1))                      bf16, written once (the weight gradients read it)
//     Z[M,D]  = dropout2(H * W2[D,F]^T + b2) + X1                              bf16 (kept for backward)
//     X[M,D]  = gamma * (Z - mean) * rstd + beta ; mean, rstd fp32 [M]         (norm2, eps inside the square root)
//
// = PositionwiseFeedForward.forward (V/models/layers/position_wise_feed_forward.py:12-17: Linear, ReLU, Dropout, Linear)
// followed by `x = norm2(dropout2(ffn(x)) + x)` of EncoderLayer.forward (V/models/blocks/encoder_layer.py:30-33); optional stages
// in front (attention output projection + dropout + residual + norm1, encoder_layer.py:24-28) and behind (the NEXT layer's packed
// q,k,v projection, multi_head_attention.py:17-19); the backward kernel mirrors it (see there).
//
// Run as two launches (FFN1 GEMM, then FFN2 GEMM + LayerNorm) the hidden activation H -- 4x the width of every other
// activation -- is written to HBM and read straight back (cfg B: 77.5 MB each way per layer), and each launch pays its own
// pipeline fill and drain.  Here the ROWS are stationary and the WEIGHTS stream:
//   * a wave owns 32 consecutive rows (two 16-row groups; 16 rows where that leaves CUs idle: chain_shape) for the whole kernel.
//     Its X1 rows sit in registers as MFMA fragments (loaded once, fragment-shaped, straight from global memory), its Z
//     accumulators (32 rows x D, fp32) too;
//   * the hidden dimension is walked in chunks of 64 units.  Per chunk: acc1 = X1 W1_c^T (K = D), epilogue (bias, ReLU,
//     Philox dropout, round to bf16, 16-byte store of H) -- and the eight bf16 values a lane has just packed ARE its
//     activation fragment of the second product (an accumulator tile as the next MFMA's operand: the k-slot order inside an
//     MFMA is free as long as both operands agree, so lane group g carries hidden units {0,16,8,24}[g] .. +7 of a 32-unit
//     step and the W2 fragment is read with the same permutation): acc2 += H_c W2_c^T without H touching LDS or HBM again;
//   * LDS holds nothing but the weight ring: chunk c's W1 rows [64 x D] and W2 columns [D x 64] (24.6 KB each at D = 192)
//     arrive by global_load_lds two chunks ahead (3 slots); on the source side the 16-byte chunks of a row are put in
//     lane-group order (the {0,16,8,24} permutation) and XOR-swizzled, so that the fragment reads (ds_read_b128, 16 rows x
//     the consecutive chunks of two lane groups) are bank-conflict free.  Every CU streams the layer's 0.6 MB of
//     FFN weights from L2 once per 224 rows.  One workgroup barrier per chunk (ring hand-over), none for the data path;
//   * tail: z = dropout2(acc2 + b2) + x1 with x1 taken from the X1 fragments (the same column permutation makes them exactly
//     the 8 columns a lane owns after the tail's permlane swap), LayerNorm statistics two-pass on the bf16-rounded z by lane
//     sums + two shuffles (a wave owns whole rows), Z / X / statistics stored from registers.
// The chunks run in ascending order, but inside a 32-unit MFMA step the k-slots are permuted: H and Z equal iq_gemm_bf16_nt +
// iq_gemm_bf16_ln up to fp32 summation order inside one MFMA, i.e. at bf16 rounding ties (tests/test_gpu_kernels.py bounds the
// fraction).  Measured, cfg B layer: 89.6 us for the whole forward launch against 126 for the four it replaces; it is bound by
// the vector ALU (per chunk and wave 601 instructions, half of them Philox, against 96 MFMAs), DESIGN.md section 6.
// First version of this file (round 3, measured, replaced; kept as scripts/dbg/variants/ffn_chain_v1_frame_images.hip): one
// workgroup per frame with X1 and H_c as LDS images and weight fragments loaded from L2 into registers 1-2 k-steps ahead: 79 us
// for cfg B's FFN against 38 + 37 for the two launches -- every k-step exposed an L2 round trip, and registers left no room
// to look further ahead.  A ring in LDS costs no registers.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int FC_CHUNK = 64, FC_NS = 3, FC_MAXW = 7;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

struct FfnChainParams {
  const bf16* X1; const bf16* W1; const bf16* W2;      // [M,D], [F,D], [D,F]
  const float* b1; const float* b2; const float* gamma; const float* beta;
  bf16* H; bf16* Z; bf16* X; float* mean; float* rstd;  // [M,F], [M,D], [M,D], [M], [M]
  uint32_t* gate;                                       // optional: "H > 0" bits for iq_ffn_chain_bwd, [ceil(M/32)][F/64][64] dwords
  int M, F;
  float eps;
  int drop1_on, drop2_on; IqRng rng1, rng2; uint32_t thresh1, thresh2; float dscale1, dscale2;
  // optional first stage (PRE): X1 = norm1(dropout0(A0 * W0[D,D]^T + b0) + R0) is computed here (and written, with Z0 / mean0 /
  // rstd0, for the backward pass) instead of read
  const bf16* A0; const bf16* W0; const bf16* R0; const float* b0; const float* gamma0; const float* beta0;
  bf16* Z0; bf16* X1out; float* mean0; float* rstd0;
  int drop0_on; IqRng rng0; uint32_t thresh0; float dscale0;
  // optional last stage (POST): Yq[M,3D] = X * Wq[3D,D]^T + bq -- the NEXT layer's packed q,k,v projection of this layer's output
  const bf16* Wq; const float* bq; bf16* Yq;
};

// W1 image rows are D * 2 bytes (24 | 16 chunks of 16 B), W2 image rows 128 B (8 chunks).  Swizzles (involutions on the chunk
// index) that put the 16 rows x {k-chunk a, b} of a ds_read_b128 lane group on 16 distinct 16-byte slots of the 256-byte bank row:
//   384 B rows (24 = 8 mod 16: the row's parity already moves the slot by 8): low 3 bits ^= (row >> 1) & 7
//   256 B rows: chunk ^= row & 15          128 B rows: chunk ^= (row >> 1) & 7
template <int CPR> __device__ __forceinline__ int fc_swz(int row, int ch) {
  return CPR == 16 ? (ch ^ (row & 15)) : CPR == 8 ? (ch ^ ((row >> 1) & 7)) : ((ch & ~7) | ((ch & 7) ^ ((row >> 1) & 7)));
}
// Weight-fragment reads are inline asm with COUNTED lgkmcnt waits: for builtin LDS reads in these loops hipcc waits
// lgkmcnt(0) before every MFMA group, i.e. also for the fragments it has just requested for the NEXT group (the read-ahead
// bought nothing: 60.9 vs 65.9 us; without MFMAs, stores and ring the loop still took 1.7 us per chunk, all of it LDS
// latency).  LDS returns in order: "at most N outstanding" retires everything older.  OFF = compile-time byte offset.
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
#define FC_LGKM_WAIT(N)                                  \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)

// FC_SGPR_HAZARD: the inline-asm memory instructions below take a wave-uniform base address in scalar registers ("s" operand).
// When the compiler has spilled that value to a VGPR lane it restores it with v_readlane_b32 right in front of the asm -- and a
// vector-memory instruction that reads an SGPR written by a VALU instruction needs five wait states (gfx9 data hazard), which the
// hazard recogniser inserts for its own instructions but not in front of inline asm: the access then used a stale address (two
// memory-access faults on the GPU, both in instances with dozens of spilled scalar registers; found in the ISA as
// `v_readlane_b32 s4, v255, 36` / `s5` directly followed by the asm's `global_load_dword v222, v148, s[4:5]`).  Every such asm
// starts with `s_nop 4`; tests/test_host_cpu.py checks that in the compiled ISA.
// hidden-unit / column offset (in units of 8) of lane group g inside a 32-wide k-step: {0, 16, 8, 24} / 8
__device__ __forceinline__ int fc_kperm(int g) { return ((g & 1) << 1) | (g >> 1); }

// NW = waves per workgroup (compile time: the ring's counted waits need the number of DMA pieces per wave), DROP1 = dropout
// on the hidden activation (its Philox rounds are the kernel's largest block of vector work; computing a chunk's keep flags
// one chunk ahead, beside the second product's MFMAs, measured slower: 71.8 vs 65.9 us).
// PRE = the attention output projection + dropout + residual + norm1 (EncoderLayer.forward, encoder_layer.py:24-28) in front: the
// projection is one more "first product" (K = D, W0's rows in 64-row blocks through the LDS ring region before the FFN weights
// need it), and after the permlane swap its output tile is, lane for lane, the activation fragment of the FFN's first product:
// norm1's output never leaves the registers on its way into the FFN.
// POST = the NEXT layer's q,k,v projection (scale_dot_product_attention's inputs, multi_head_attention.py:17-19) behind norm2:
// its weight's 64-row blocks are further ring "chunks" (two blocks fill one slot exactly), norm2's output tile is their activation
// fragment, the tail runs between the last FFN chunk and the first of them while they are already in flight.
// MODE 0: the feed-forward sub-layer alone; 1: PRE; 2: PRE and POST.
template <int D, int NW, int RG, bool DROP1, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void ffn_chain_fwd_kernel(const FfnChainParams p) {
  constexpr int RW = 16 * RG;                           // rows per wave: RG 16-row groups (2; 1 where 32-row waves would leave CUs idle)
  constexpr bool PRE = MODE >= 1, POST = MODE == 2;
  constexpr int XCPR = D / 8;                           // 16-byte chunks per W1 image row
  constexpr int KS1 = D / 32;                           // k-steps of the first product: 6 | 4
  constexpr int NT2 = D / 16, NP2 = NT2 / 2;            // output column tiles / pairs: 12, 6 | 8, 4
  constexpr int W1_BYTES = FC_CHUNK * D * 2, W2_BYTES = D * FC_CHUNK * 2, SLOT = W1_BYTES + W2_BYTES;
  constexpr int W1_PIECES = W1_BYTES / 1024, PIECES = SLOT / 1024;     // 1 KiB DMA pieces per chunk: 48 | 32
  constexpr int PPW = (PIECES + NW - 1) / NW;                          // per wave (a wave past the end repeats the last piece)
  static_assert(PPW + 8 * RG < 64, "counted vmcnt");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwave = NW;
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  const int F = p.F;
  const int nchunk = F / FC_CHUNK;
  const long row0 = ((long)blockIdx.x * nwave + wave) * RW;      // this wave's first row
  const bool have = row0 < p.M;                                   // wave-uniform (a trailing wave may own no rows: it still feeds the ring)

  // ---- weight ring: this wave's PPW pieces of every chunk; per-lane source offsets are chunk-invariant ------------------------
  // image position (row r, slot s) holds "slot-order" chunk q = swz(r, s), i.e. the k-chunk lane group (q & 3) reads in
  // k-step (q >> 2): natural chunk 4 (q >> 2) + kperm(q & 3)
  // (per-lane source offsets are recomputed per chunk: a dozen integer instructions per piece against seven resident registers)
  auto issue_chunk = [&](int c) {
    unsigned char* slot = smem + (c % FC_NS) * SLOT;
    const char* base1 = reinterpret_cast<const char*>(p.W1 + (long)c * FC_CHUNK * D);
    const char* base2 = reinterpret_cast<const char*>(p.W2 + (long)c * FC_CHUNK);
    int ln = lane;
    asm volatile("" : "+v"(ln));                        // (keeps the offsets below from being hoisted out of the chunk loop)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = min(wave + i * NW, PIECES - 1);    // wave-uniform
      const int lin = pc * 64 + ln;
      unsigned off;
      if (pc < W1_PIECES) {                             // W1 rows f0 .. f0 + 63, all D columns
        const int r = lin / XCPR, sl = lin - r * XCPR;
        const int q = fc_swz<XCPR>(r, sl);
        off = (unsigned)((r * D + ((q & ~3) | fc_kperm(q & 3)) * 8) * 2);
      } else {                                          // W2 rows 0 .. D - 1, columns f0 .. f0 + 63
        const int l2 = lin - W1_PIECES * 64;
        const int r = l2 >> 3, sl = l2 & 7;
        const int q = fc_swz<8>(r, sl);
        off = (unsigned)((r * F + ((q & ~3) | fc_kperm(q & 3)) * 8) * 2);
      }
      const char* base = pc < W1_PIECES ? base1 : base2;
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(base + off), (lds_void_t*)(slot + pc * 1024), 16, 0, 0);
    }
  };
  // POST: q,k,v weight rows 128 q .. 128 q + 127 as two W1-type images (one slot); rows past 3 D (the last, half chunk) repeat the
  // last row (never read)
  constexpr int NBQ = 3 * D / FC_CHUNK, NQ = (NBQ + 1) / 2;      // 9 blocks in 5 chunks | 6 in 3
  static_assert(2 * W1_BYTES == SLOT, "two projection blocks fill a ring slot");
  const int ntot = nchunk + (POST ? NQ : 0);
  auto issue_any = [&](int c) {
    if (!POST || c < nchunk) { issue_chunk(c); return; }
    const int q = c - nchunk;
    unsigned char* slot = smem + (c % FC_NS) * SLOT;
    int ln = lane;
    asm volatile("" : "+v"(ln));                        // (not loop-invariant as far as the compiler knows: see issue_chunk)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = min(wave + i * NW, PIECES - 1);
      const int lin = pc * 64 + ln;
      const int r = lin / XCPR, sl = lin - r * XCPR;
      const int qq = fc_swz<XCPR>(r & 63, sl);
      const int grow = min(q * 2 * FC_CHUNK + r, 3 * D - 1);
      const unsigned off = (unsigned)((grow * D + ((qq & ~3) | fc_kperm(qq & 3)) * 8) * 2);
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(reinterpret_cast<const char*>(p.Wq) + off), (lds_void_t*)(slot + pc * 1024), 16, 0, 0);
    }
  };
  // PRE: W0's NB0 row blocks (W1-type images, contiguous from ring slot 1 on: 73,728 of 98,304 | 32,768 of 32,768 bytes); the
  // FFN's chunk 0 travels beside them into slot 0, chunk 1 follows once the projection has been consumed
  constexpr int NB0 = D / FC_CHUNK;
  if (PRE) {
    constexpr int PRE_PIECES = NB0 * W1_PIECES, PRE_PPW = (PRE_PIECES + NW - 1) / NW;
    static_assert(NB0 * W1_BYTES <= 2 * SLOT, "projection weight blocks fit ring slots 1 and 2");
#pragma unroll
    for (int i = 0; i < PRE_PPW; ++i) {
      const int pc = min(wave + i * NW, PRE_PIECES - 1);
      const int lin = pc * 64 + lane;                   // (blocks are contiguous in both the image and, row-wise, in W0)
      const int r = lin / XCPR, sl = lin - r * XCPR;
      const int q = fc_swz<XCPR>(r & 63, sl);
      const unsigned off = (unsigned)((r * D + ((q & ~3) | fc_kperm(q & 3)) * 8) * 2);
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(reinterpret_cast<const char*>(p.W0) + off),
                                       (lds_void_t*)(smem + SLOT + pc * 1024), 16, 0, 0);
    }
  }
  issue_chunk(0);
  if (!PRE && ntot > 1) issue_any(1);

  // ---- this wave's X1 rows as activation fragments: xf[rg][ks] = row 16 rg + c16, columns 32 ks + 8 kperm(g) .. +7 ----------
  // (PRE: the attention output's rows first -- the projection's activation fragments -- then norm1's output in the same registers)
  bf16x8 xf[RG][KS1];
  {
    const bf16* src = PRE ? p.A0 : p.X1;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const long r = min(row0 + rg * 16 + c16, (long)p.M - 1);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xf[rg][ks] = have ? *reinterpret_cast<const bf16x8*>(src + r * D + 32 * ks + 8 * fc_kperm(g)) : bf16x8{};
    }
  }
  // PRE: the residual rows too, in the projection's output layout (pair jp: columns 32 jp + 8 kperm(g) .. +7, as xf) -- fetched per
  // column pair inside the projection loop, each 200-cycle MFMA group waited for an HBM round trip (23 us for the stage)
  bf16x8 res0[PRE ? RG : 1][PRE ? KS1 : 1];
  if (PRE) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const long r = min(row0 + rg * 16 + c16, (long)p.M - 1);
#pragma unroll
      for (int jp = 0; jp < KS1; ++jp)
        res0[rg][jp] = have ? *reinterpret_cast<const bf16x8*>(p.R0 + r * D + 32 * jp + 8 * fc_kperm(g)) : bf16x8{};
    }
  }
  const IqRng rng1 = DROP1 ? rng_resolve(p.rng1) : p.rng1;
  const IqRng rng2 = p.drop2_on ? rng_resolve(p.rng2) : p.rng2;
  // b1 lives in LDS behind the ring: a register-destination global load inside the chunk loop makes hipcc wait vmcnt(0) at
  // its first use (cdna_hip_programming.md, "mixing load kinds in one k-loop"), i.e. for the ring pieces just requested.
  // Behind it the per-column vectors of the tail (b2, gamma, beta) and of the first stage (b0, gamma0, beta0): [6][D] floats.
  float* b1s = reinterpret_cast<float*>(smem + FC_NS * SLOT);
  float* vecs = b1s + F;
  for (int i = tid; i < F / 4; i += NW * 64) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(p.b1)[i];
  for (int i = tid; i < (PRE ? 6 : 3) * (D / 4); i += NW * 64) {
    const int v = i / (D / 4), j = i - v * (D / 4);
    const float* src = v == 0 ? p.b2 : v == 1 ? p.gamma : v == 2 ? p.beta : v == 3 ? p.b0 : v == 4 ? p.gamma0 : p.beta0;
    reinterpret_cast<f32x4*>(vecs)[i] = reinterpret_cast<const f32x4*>(src)[j];
  }
  if (POST)                                             // bq [3 D] behind them
    for (int i = tid; i < 3 * D / 4; i += NW * 64) reinterpret_cast<f32x4*>(vecs + 6 * D)[i] = reinterpret_cast<const f32x4*>(p.bq)[i];
  // Everything requested so far has landed (chunks 0 and 1 as well: they had the X1 round trip to travel); the X1 fragments
  // are then passed through an empty asm so that the compiler stops tracking them as results of pending loads -- it cannot
  // see across the loop's back edge and would otherwise wait vmcnt(0) before their first use in EVERY iteration.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(xf[rg][ks]));
  if (PRE) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int jp = 0; jp < KS1; ++jp) asm volatile("" : "+v"(res0[rg][jp]));
  }

  // slot-relative byte offsets of this lane's fragment reads: row c16 of a 16-row tile (tiles add a compile-time offset: the
  // swizzle only looks at (row >> 1) & 7 = (c16 >> 1) & 7), k-step ks of the W1 image / jp of the W2 image
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  // (384-byte rows: the swizzle leaves the chunk's bits above 3 alone, k-step ks reads 128 (ks >> 1) bytes behind k-step ks & 1:
  //  two resident offsets instead of six)
  constexpr int W1_KMASK = XCPR == 24 ? 1 : KS1 - 1;
  uint32_t w1off[KS1], w2off[2];
  auto w1at = [&](int ks) -> uint32_t { return w1off[ks & W1_KMASK] + (XCPR == 24 ? 128u * (uint32_t)(ks >> 1) : 0u); };
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) w1off[ks] = (uint32_t)(c16 * (D * 2) + fc_swz<XCPR>(c16, 4 * (ks & W1_KMASK) + g) * 16);
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) w2off[jp] = (uint32_t)(W1_BYTES + c16 * 128 + fc_swz<8>(c16, 4 * jp + g) * 16);

  if (PRE) {
    __syncthreads();                                    // every wave's pieces of W0 are in LDS
    if (have) {
      const IqRng rng0 = p.drop0_on ? rng_resolve(p.rng0) : p.rng0;
      bf16x8 zb[RG][KS1];                                // z0 = dropout0(A0 W0^T + b0) + R0, bf16 as stored, [rg][32-column pair]
#pragma unroll
      for (int b = 0; b < NB0; ++b) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = 32 * (2 * b + jp) + (odd ? 16 + 4 * (g - 1) : 4 * g);
          f32x4 acc1[RG][2];
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) { acc1[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          bf16x8 wp[2][2];
          const uint32_t blk = lds0 + SLOT + b * W1_BYTES;
          auto read_w0 = [&](int ks, bf16x8 (&dst)[2]) {
            const uint32_t a0 = blk + w1at(ks);
            if (jp == 0) { lds_read128<0 * 16 * D * 2>(dst[0], a0); lds_read128<1 * 16 * D * 2>(dst[1], a0); }
            else { lds_read128<2 * 16 * D * 2>(dst[0], a0); lds_read128<3 * 16 * D * 2>(dst[1], a0); }
          };
          read_w0(0, wp[0]);
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks) {
            if (ks + 1 < KS1) { read_w0(ks + 1, wp[(ks + 1) & 1]); FC_LGKM_WAIT(2); }
            else FC_LGKM_WAIT(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
              for (int rg = 0; rg < RG; ++rg)
                acc1[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks & 1][t], xf[rg][ks], acc1[rg][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          const f32x4 b_lo = *reinterpret_cast<const f32x4*>(vecs + 3 * D + col), b_hi = *reinterpret_cast<const f32x4*>(vecs + 3 * D + col + 4);
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) {
            float w[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float va = acc1[rg][0][r], vb = acc1[rg][1][r];
              const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
              w[r] = __uint_as_float(sw[0]) + b_lo[r];
              w[4 + r] = __uint_as_float(sw[1]) + b_hi[r];
            }
            const long grow = row0 + rg * 16 + c16;
            if (p.drop0_on) {
              const uint32_t keep = dropout_keep8(rng0, (uint64_t)(grow * D + col) >> 3, p.thresh0);
#pragma unroll
              for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale0 : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] += (float)res0[rg][2 * b + jp][e];
            zb[rg][2 * b + jp] = pack8(w);
            if (grow < p.M) *reinterpret_cast<bf16x8*>(p.Z0 + grow * D + col) = zb[rg][2 * b + jp];
          }
        }
      }
      // norm1 on the wave's own rows (two-pass on the bf16-rounded z0, as the tail below); its output is the FFN's fragment
      const float invD0 = 1.0f / (float)D;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        const long grow = row0 + rg * 16 + c16;
        float s1 = 0.f;
#pragma unroll
        for (int jp = 0; jp < KS1; ++jp)
#pragma unroll
          for (int e = 0; e < 8; ++e) s1 += (float)zb[rg][jp][e];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 * invD0;
        float s2 = 0.f;
#pragma unroll
        for (int jp = 0; jp < KS1; ++jp)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = (float)zb[rg][jp][e] - mean; s2 += d * d; }
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = 1.0f / sqrtf(s2 * invD0 + p.eps);
        if (g == 0 && grow < p.M) { p.mean0[grow] = mean; p.rstd0[grow] = rstd; }
#pragma unroll
        for (int jp = 0; jp < KS1; ++jp) {
          const int col = 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);
          const f32x4 g_lo = *reinterpret_cast<const f32x4*>(vecs + 4 * D + col), g_hi = *reinterpret_cast<const f32x4*>(vecs + 4 * D + col + 4);
          const f32x4 e_lo = *reinterpret_cast<const f32x4*>(vecs + 5 * D + col), e_hi = *reinterpret_cast<const f32x4*>(vecs + 5 * D + col + 4);
          float y[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = g_lo[e] * (((float)zb[rg][jp][e] - mean) * rstd) + e_lo[e];
            y[4 + e] = g_hi[e] * (((float)zb[rg][jp][4 + e] - mean) * rstd) + e_hi[e];
          }
          xf[rg][jp] = pack8(y);
          if (grow < p.M) *reinterpret_cast<bf16x8*>(p.X1out + grow * D + col) = xf[rg][jp];
        }
      }
    }
    // every wave is done with W0's image: the ring proper starts (the stage's stores stay in flight: they are OLDER than any
    // ring piece requested from here on, the loop's counted waits retire them first)
    __syncthreads();
    if (ntot > 1) issue_any(1);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(xf[rg][ks]));
  }

  f32x4 acc2[RG][NT2];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int j = 0; j < NT2; ++j) acc2[rg][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // a wave whose 32 rows all exist stores exactly 4 H vectors per chunk: its ring waits can be COUNTED (the stores and the
  // next chunk's pieces stay in flight); a ragged wave waits for everything
  const bool full = row0 + RW <= p.M;
  const bool gated = p.gate != nullptr;                 // (kernel-uniform) one more store per chunk
  // (Dropout of the hidden activation: four Philox calls per lane and chunk, ~1,800 cycles of integer multiplies, 13-18 of the
  //  kernel's ~62 us.  Computing them one chunk ahead -- all four before the second product, or one call after each of four
  //  MFMA groups "in the shadow of the matrix pipe" -- measured SLOWER both times (71.8 and 68.5 against 63.6 us): the vector
  //  ALU is the shared resource, MFMA issue needs its slots too.  They stay in the epilogue, where their result is used.)
  for (int c = 0; c < nchunk; ++c) {
    // chunk c has landed for this wave (counted: the chunk requested one iteration ago and the H stores of the last two
    // iterations may still be in flight); the barrier makes that true for every wave and says slot (c+2) % 3 = (c-1) % 3 is
    // no longer read
#ifdef FC_NO_RING      // ablation build (timing only): no ring hand-over after the prologue
    if (c > 0) goto ring_done;
#endif
    if (c == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (b1 image written; the ring's first chunks: above)
    else if (!full || nchunk < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (gated && c + 1 < ntot) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW + 4 * RG + 2) : "memory");
    else if (gated) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * RG + 2) : "memory");
    else if (c + 1 < ntot) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW + 4 * RG) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * RG) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 2 < ntot) issue_any(c + 2);
#ifdef FC_NO_RING
  ring_done:
#endif
    if (!have) continue;
    const int f0 = c * FC_CHUNK;
    // ---- first product + its epilogue, 32 hidden units (two column tiles) at a time: 16 accumulator registers and a 16-register
    //      read-ahead instead of 32 + 32 (the wide form left no room for the Philox temporaries between the MFMA groups).  The
    //      packed values are the second product's activation fragments. -------------------------------------------------------
    const uint32_t slot_addr = lds0 + (c % FC_NS) * SLOT;
    bf16x8 hf[RG][2];                                    // [rg][k-step of the chunk]
    uint32_t gbits = 0;                                 // "hidden unit > 0" (= ReLU and dropout gate of the backward), byte 2 jp + rg
    bf16x8 wp[2][2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x4 acc1[RG][2];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) { acc1[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      auto read_w1 = [&](int ks, bf16x8 (&dst)[2]) {
        const uint32_t a0 = slot_addr + w1at(ks);
        if (jp == 0) { lds_read128<0 * 16 * D * 2>(dst[0], a0); lds_read128<1 * 16 * D * 2>(dst[1], a0); }
        else { lds_read128<2 * 16 * D * 2>(dst[0], a0); lds_read128<3 * 16 * D * 2>(dst[1], a0); }
      };
      read_w1(0, wp[0]);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) { read_w1(ks + 1, wp[(ks + 1) & 1]); FC_LGKM_WAIT(2); }
        else FC_LGKM_WAIT(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#ifdef FC_NO_MFMA      // ablation build (timing only)
          asm volatile("" :: "v"(wp[ks & 1][t]));
#else
#pragma unroll
          for (int rg = 0; rg < RG; ++rg)
            acc1[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks & 1][t], xf[rg][ks], acc1[rg][t], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int col = f0 + 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);       // first of this lane's 8 hidden units
      const f32x4 b_lo = *reinterpret_cast<const f32x4*>(b1s + col), b_hi = *reinterpret_cast<const f32x4*>(b1s + col + 4);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        float w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float va = acc1[rg][0][r], vb = acc1[rg][1][r];
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
          w[r] = fmaxf(__uint_as_float(sw[0]) + b_lo[r], 0.f);
          w[4 + r] = fmaxf(__uint_as_float(sw[1]) + b_hi[r], 0.f);
        }
        const long grow = row0 + rg * 16 + c16;
        if (DROP1) {
          const uint32_t keep = dropout_keep8(rng1, (uint64_t)(grow * F + col) >> 3, p.thresh1);
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale1 : 0.f;
        }
        hf[rg][jp] = pack8(w);
#pragma unroll
        for (int e = 0; e < 8; ++e) gbits |= (w[e] > 0.f ? 1u : 0u) << (8 * (2 * jp + rg) + e);
#ifndef FC_NO_HSTORE    // (ablation build, timing only, leaves H unwritten)
        if (grow < p.M) *reinterpret_cast<bf16x8*>(p.H + grow * F + col) = hf[rg][jp];
#endif
      }
    }
    // one dword per lane and chunk: 256 contiguous bytes per wave (a fifth store per chunk when the caller asks for the bits:
    // the counted waits above)
    {
      if (gated) p.gate[((row0 / RW) * nchunk + c) * 64 + lane] = gbits;
    }
    // (same read-ahead: groups of four output tiles; group q = k-step q / NG4, tiles 4 (q % NG4) .. + 3)
    constexpr int NG4 = NT2 / 4, NGRP = 2 * NG4;        // 3 | 2 groups per k-step, 6 | 4 per chunk
    bf16x8 wq[2][4];
    auto read_w2 = [&](auto qc, bf16x8 (&dst)[4]) {
      constexpr int q = decltype(qc)::value;
      constexpr int jp = q / NG4, j0 = 4 * (q % NG4);
      const uint32_t a0 = slot_addr + w2off[jp];
      lds_read128<(j0 + 0) * 16 * 128>(dst[0], a0);
      lds_read128<(j0 + 1) * 16 * 128>(dst[1], a0);
      lds_read128<(j0 + 2) * 16 * 128>(dst[2], a0);
      lds_read128<(j0 + 3) * 16 * 128>(dst[3], a0);
    };
    auto p2 = [&](auto self, auto qc) -> void {
      constexpr int q = decltype(qc)::value;
      if constexpr (q < NGRP) {
        if constexpr (q + 1 < NGRP) { read_w2(std::integral_constant<int, q + 1>{}, wq[(q + 1) & 1]); FC_LGKM_WAIT(4); }
        else FC_LGKM_WAIT(0);
        constexpr int jp = q / NG4, j0 = 4 * (q % NG4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#ifdef FC_NO_MFMA
          asm volatile("" :: "v"(wq[q & 1][t]), "v"(hf[0][jp]), "v"(hf[RG - 1][jp]));
#else
#pragma unroll
          for (int rg = 0; rg < RG; ++rg)
            acc2[rg][j0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q & 1][t], hf[rg][jp], acc2[rg][j0 + t], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        self(self, std::integral_constant<int, q + 1>{});
      }
    };
    read_w2(std::integral_constant<int, 0>{}, wq[0]);
    p2(p2, std::integral_constant<int, 0>{});
  }
  if (!POST && !have) return;

  // ---- tail: z = dropout2(acc2 + b2) + x1 (bf16), LayerNorm over the wave's own rows -----------------------------------------
  // pair jp covers columns 32 jp .. 32 jp + 31; after the swap this lane holds columns 32 jp + 8 kperm(g) .. +7 -- exactly xf[rg][jp]
  // (POST: norm2's output replaces x1 in xf -- the projection's activation fragments)
  const float invD = 1.0f / (float)D;
  if (have) {
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const long grow = row0 + rg * 16 + c16;
    float z[NP2][8];
    float s1 = 0.f;
#pragma unroll
    for (int jp = 0; jp < NP2; ++jp) {
      const int col = 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);
      const f32x4 b_lo = *reinterpret_cast<const f32x4*>(vecs + col), b_hi = *reinterpret_cast<const f32x4*>(vecs + col + 4);
      float w[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float va = acc2[rg][2 * jp][r], vb = acc2[rg][2 * jp + 1][r];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        w[r] = __uint_as_float(sw[0]) + b_lo[r];
        w[4 + r] = __uint_as_float(sw[1]) + b_hi[r];
      }
      if (p.drop2_on) {
        const uint32_t keep = dropout_keep8(rng2, (uint64_t)(grow * D + col) >> 3, p.thresh2);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale2 : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] += (float)xf[rg][jp][e];
      const bf16x8 zb = pack8(w);
      if (grow < p.M) *reinterpret_cast<bf16x8*>(p.Z + grow * D + col) = zb;
      unpack8(zb, z[jp]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s1 += z[jp][e];
    }
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * invD;
    float s2 = 0.f;
#pragma unroll
    for (int jp = 0; jp < NP2; ++jp)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = z[jp][e] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = 1.0f / sqrtf(s2 * invD + p.eps);
    if (grow < p.M) {
      if (g == 0) { p.mean[grow] = mean; p.rstd[grow] = rstd; }
#pragma unroll
      for (int jp = 0; jp < NP2; ++jp) {
        const int col = 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);
        const f32x4 g_lo = *reinterpret_cast<const f32x4*>(vecs + D + col), g_hi = *reinterpret_cast<const f32x4*>(vecs + D + col + 4);
        const f32x4 e_lo = *reinterpret_cast<const f32x4*>(vecs + 2 * D + col), e_hi = *reinterpret_cast<const f32x4*>(vecs + 2 * D + col + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = g_lo[e] * ((z[jp][e] - mean) * rstd) + e_lo[e];
          y[4 + e] = g_hi[e] * ((z[jp][4 + e] - mean) * rstd) + e_hi[e];
        }
        const bf16x8 yb = pack8(y);
        *reinterpret_cast<bf16x8*>(p.X + grow * D + col) = yb;
        if (POST) xf[rg][jp] = yb;
      }
    }
  }
  }
  if (!POST) return;

  // ---- POST: Yq = X Wq^T + bq, 64 output columns (one weight block) at a time, two blocks per ring chunk ------------------------
  for (int q = 0; q < NQ; ++q) {
    const int c = nchunk + q;
    // q = 0 drains the tail's stores with the ring (chunks q = 0 and 1 have landed then); later: younger than chunk c's pieces are
    // the 8 + 8 Yq stores of the last two iterations and the pieces of chunk c + 1
    if (q == 0 || !full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (c + 1 < ntot) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW + 8 * RG) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * RG) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 2 < ntot) issue_any(c + 2);
    if (!have) continue;
    const uint32_t slot_addr = lds0 + (c % FC_NS) * SLOT;
#pragma unroll
    for (int bl = 0; bl < 2; ++bl) {
      if (2 * q + bl >= NBQ) break;                     // (the last chunk of an odd block count is half empty)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        f32x4 acc1[RG][2];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) { acc1[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        bf16x8 wp[2][2];
        const uint32_t blk = slot_addr + bl * W1_BYTES;
        auto read_wq = [&](int ks, bf16x8 (&dst)[2]) {
          const uint32_t a0 = blk + w1at(ks);
          if (jp == 0) { lds_read128<0 * 16 * D * 2>(dst[0], a0); lds_read128<1 * 16 * D * 2>(dst[1], a0); }
          else { lds_read128<2 * 16 * D * 2>(dst[0], a0); lds_read128<3 * 16 * D * 2>(dst[1], a0); }
        };
        read_wq(0, wp[0]);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          if (ks + 1 < KS1) { read_wq(ks + 1, wp[(ks + 1) & 1]); FC_LGKM_WAIT(2); }
          else FC_LGKM_WAIT(0);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
              acc1[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks & 1][t], xf[rg][ks], acc1[rg][t], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const int col = FC_CHUNK * (2 * q + bl) + 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);
        const f32x4 b_lo = *reinterpret_cast<const f32x4*>(vecs + 6 * D + col), b_hi = *reinterpret_cast<const f32x4*>(vecs + 6 * D + col + 4);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          float w[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float va = acc1[rg][0][r], vb = acc1[rg][1][r];
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
            w[r] = __uint_as_float(sw[0]) + b_lo[r];
            w[4 + r] = __uint_as_float(sw[1]) + b_hi[r];
          }
          const long grow = row0 + rg * 16 + c16;
          if (grow < p.M) *reinterpret_cast<bf16x8*>(p.Yq + grow * (3 * D) + col) = pack8(w);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the same sub-layer's data path, one launch (autograd of position_wise_feed_forward.py:12-17 + the norm1 that
// feeds it, encoder_layer.py:24-25):
//     gH[M,F]  = (H > 0) ? (dO[M,D] * W2t[F,D]^T) * gate_scale : 0            bf16, written once (the W1 / W2 weight gradients read it)
//     dX1[M,D] = gH * W1t[D,F]^T + R                                           fp32, rounded to bf16 as the unfused GEMM stored it
//     then exactly iq_ln_bwd on it:  g = dX1 * gamma, xhat = (Z1 - mean) * rstd,
//     dZ = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),  dY = dropout1_mask(dZ) * scale,
//     partial[workgroup] = sums over its rows of (dX1 * xhat | dX1)            fp32 [2 D]  (norm1's gamma / beta gradient partials)
// replacing the gate data-gradient GEMM and the FFN1 data-gradient GEMM + LayerNorm backward (gemm_lnbwd.hip): gH no longer
// makes the round trip through HBM between them.  Same structure as the forward kernel above: dO rows as register
// fragments, the two transposed weights through the LDS ring, the gated tile handed to the second product in registers.
// The gate ("H > 0": ReLU and dropout1 of the forward pass at once) is one BIT per hidden unit, written by the forward kernel
// in this kernel's own wave / chunk / lane order: one dword per lane and chunk instead of four 16-byte rows of H (H itself is
// read by the weight gradients only; 77.5 MB per layer less for cfg B), and one register instead of sixteen -- with the gate
// rows resident beside the dO fragments and the accumulators the D = 192 build spilled 150 registers.
struct FfnChainBwdParams {
  const bf16* dO; const bf16* W2t; const bf16* W1t;     // [M,D], [F,D], [D,F]
  const uint32_t* gate; const bf16* R; const bf16* Z1;  // gate bits of the forward kernel, [M,D], [M,D]
  const float* mean; const float* rstd; const float* gamma;
  bf16* gH; bf16* dZ; bf16* dY; float* partial;         // [M,F], [M,D], [M,D], [workgroups][2 D]
  int M, F;
  float gate_scale;
  int drop_on; IqRng rng; uint32_t thresh; float dscale;
  // optional last stage (POSTB): dA[M,D] = dY * Wot[D,D]^T -- the attention output projection's data gradient (Wot = Wo transposed)
  const bf16* Wot; bf16* dA;
  // optional first stage (PREB): dO is not read but computed -- dX2 = A0[M,3D] * W0t[D,3D]^T + R0, norm2 backward on it with Z0 /
  // mean0 / rstd0 / gamma0 (dropout site rng0, same probability): dZ0, dY0 (= dO) and partial0 are written
  const bf16* A0; const bf16* W0t; const bf16* R0; const bf16* Z0; const float* mean0; const float* rstd0; const float* gamma0;
  IqRng rng0; bf16* dZ0; bf16* dY0; float* partial0;
};

// MODE 1, 2 (POSTB): the tail's dY rows go back into the wave's LDS image, are re-read as MFMA activation fragments and multiplied
// with the transposed projection weight (64-row blocks in the ring slots the loop has left): dA = dY Wo, the gradient the
// attention backward starts from, without dY's round trip through HBM and one launch less.
// MODE 2 (PREB): in front, the data gradient of the q,k,v projection of the layer ABOVE + this layer's norm2 backward (what
// gemm_lnbwd.hip does as a launch of its own): dX2 = A0[M,3D] * W0t[D,3D]^T + R0 is one more "second product" -- K = 3 D walked
// in 128-column chunk pairs through the same ring, the A0 rows fetched as fragments one pair ahead (inline-asm loads behind
// counted waits) -- followed by the same LayerNorm-backward tail; its dY rows are this kernel's dO fragments (they never leave
// the CU on their way into the feed-forward backward), dZ0 / dY0 / partial0 are written for the weight gradients and the tail.
template <int D, int NW, int RG, bool DROP, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void ffn_chain_bwd_kernel(const FfnChainBwdParams p) {
  constexpr int RW = 16 * RG;                           // rows per wave
  constexpr bool POSTB = MODE >= 1, PREB = MODE == 2;
  constexpr int XCPR = D / 8;
  constexpr int KS1 = D / 32;
  constexpr int NT2 = D / 16, NP2 = NT2 / 2;
  constexpr int W1_BYTES = FC_CHUNK * D * 2, W2_BYTES = D * FC_CHUNK * 2, SLOT = W1_BYTES + W2_BYTES;
  constexpr int W1_PIECES = W1_BYTES / 1024, PIECES = SLOT / 1024;
  constexpr int PPW = (PIECES + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  const int F = p.F;
  const int nchunk = F / FC_CHUNK;
  const long row0 = ((long)blockIdx.x * NW + wave) * RW;
  const bool have = row0 < p.M;
  const bool full = row0 + RW <= p.M;

  // ---- weight ring (as in the forward kernel: "W1" = W2t rows f0 .. f0+63 x D, "W2" = W1t rows 0 .. D-1 x hidden f0 .. f0+63) -----
  // (per-lane source offsets recomputed per chunk: seven resident registers more made the D = 192 build spill)
  auto issue_chunk = [&](int c) {
    unsigned char* slot = smem + (c % FC_NS) * SLOT;
    const char* base1 = reinterpret_cast<const char*>(p.W2t + (long)c * FC_CHUNK * D);
    const char* base2 = reinterpret_cast<const char*>(p.W1t + (long)c * FC_CHUNK);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = min(wave + i * NW, PIECES - 1);
      const int lin = pc * 64 + lane;
      unsigned off;
      if (pc < W1_PIECES) {
        const int r = lin / XCPR, sl = lin - r * XCPR;
        const int q = fc_swz<XCPR>(r, sl);
        off = (unsigned)((r * D + ((q & ~3) | fc_kperm(q & 3)) * 8) * 2);
      } else {
        const int l2 = lin - W1_PIECES * 64;
        const int r = l2 >> 3, sl = l2 & 7;
        const int q = fc_swz<8>(r, sl);
        off = (unsigned)((r * F + ((q & ~3) | fc_kperm(q & 3)) * 8) * 2);
      }
      const char* base = pc < W1_PIECES ? base1 : base2;
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(base + off), (lds_void_t*)(slot + pc * 1024), 16, 0, 0);
    }
  };
  long rowc[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) rowc[rg] = min(row0 + rg * 16 + c16, (long)p.M - 1);
  // slot-relative byte offsets of this lane's fragment reads (as in the forward kernel)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  // (384-byte rows: the swizzle leaves the chunk's bits above 3 alone, k-step ks reads 128 (ks >> 1) bytes behind k-step ks & 1:
  //  two resident offsets instead of six)
  constexpr int W1_KMASK = XCPR == 24 ? 1 : KS1 - 1;
  uint32_t w1off[KS1], w2off[2];
  auto w1at = [&](int ks) -> uint32_t { return w1off[ks & W1_KMASK] + (XCPR == 24 ? 128u * (uint32_t)(ks >> 1) : 0u); };
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) w1off[ks] = (uint32_t)(c16 * (D * 2) + fc_swz<XCPR>(c16, 4 * (ks & W1_KMASK) + g) * 16);
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) w2off[jp] = (uint32_t)(W1_BYTES + c16 * 128 + fc_swz<8>(c16, 4 * jp + g) * 16);

  bf16x8 xf[RG][KS1];
  f32x4 acc2[RG][NT2];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int j = 0; j < NT2; ++j) acc2[rg][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- acc2 += h (two 32-unit k-steps of this wave's rows, as fragments) x the W2-type image at img2 + W1_BYTES: model columns
  //      16 j .. +15, units 0 .. 63 (read-ahead in groups of four output tiles; group q = k-step q / NG4, tiles 4 (q % NG4) .. + 3)
  auto second_product = [&](uint32_t img2, bf16x8 (&h)[RG][2]) {
    constexpr int NG4 = NT2 / 4, NGRP = 2 * NG4;
    bf16x8 wq[2][4];
    auto read_w2 = [&](auto qc, bf16x8 (&dst)[4]) {
      constexpr int q = decltype(qc)::value;
      constexpr int jp = q / NG4, j0 = 4 * (q % NG4);
      const uint32_t a0 = img2 + w2off[jp];
      lds_read128<(j0 + 0) * 16 * 128>(dst[0], a0);
      lds_read128<(j0 + 1) * 16 * 128>(dst[1], a0);
      lds_read128<(j0 + 2) * 16 * 128>(dst[2], a0);
      lds_read128<(j0 + 3) * 16 * 128>(dst[3], a0);
    };
    auto p2 = [&](auto self, auto qc) -> void {
      constexpr int q = decltype(qc)::value;
      if constexpr (q < NGRP) {
        if constexpr (q + 1 < NGRP) { read_w2(std::integral_constant<int, q + 1>{}, wq[(q + 1) & 1]); FC_LGKM_WAIT(4); }
        else FC_LGKM_WAIT(0);
        constexpr int jp = q / NG4, j0 = 4 * (q % NG4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int rg = 0; rg < RG; ++rg)
            acc2[rg][j0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q & 1][t], h[rg][jp], acc2[rg][j0 + t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        self(self, std::integral_constant<int, q + 1>{});
      }
    };
    read_w2(std::integral_constant<int, 0>{}, wq[0]);
    p2(p2, std::integral_constant<int, 0>{});
  };

  // ---- LayerNorm-backward tail: dX = acc2 + R, rounded to bf16 as the unfused data-gradient GEMM stored it, into a wave-private
  //      LDS image; then ln_bwd_kernel's own arithmetic on it in LayerNorm's layout (a row = 8 lanes x NV 16-byte vectors): the
  //      accumulators are released at once (doing the LayerNorm in the MFMA register layout -- 96 accumulators + the row's Z and
  //      dX -- spilled ~80 registers at D = 192, and the allocator then also spilled loop-invariant fragments).
  //      reload: dY goes back into the image and is re-read as the activation fragments xf of the product that follows.
  constexpr int LDI = D;                                // image row (elements; unpadded: exactly four images per ring slot)
  constexpr int IMG = RW * LDI * 2;                     // bytes per wave: 12,288 | 8,192 (RG = 2)
  static_assert(SLOT % IMG == 0 && NW <= 2 * (SLOT / IMG), "whole wave images, all of them in two ring slots");
  auto ln_tail = [&](const bf16* Rp, const bf16* Zp, const float* meanp, const float* rstdp, const float* gammap, const IqRng& rngx,
                     uint32_t thr, float dsc, bf16* dZp, bf16* dYp, bf16* img, float* wsp, auto reload_c) {
    constexpr bool reload = decltype(reload_c)::value;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const long gr = min(row0 + rg * 16 + c16, (long)p.M - 1);
  #pragma unroll
      for (int jp = 0; jp < NP2; ++jp) {
        const int col = 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);
        const bf16x8 res = *reinterpret_cast<const bf16x8*>(Rp + gr * D + col);
        float w[8];
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float va = acc2[rg][2 * jp][r], vb = acc2[rg][2 * jp + 1][r];
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
          w[r] = __uint_as_float(sw[0]);
          w[4 + r] = __uint_as_float(sw[1]);
        }
  #pragma unroll
        for (int e = 0; e < 8; ++e) w[e] += (float)res[e];
        *reinterpret_cast<bf16x8*>(img + (rg * 16 + c16) * LDI + col) = pack8(w);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private image: the wave's own LDS writes, no barrier
    {
      constexpr int LPR = 8, NV = D / (8 * LPR);          // 3 | 2
      const int lj = lane & 7, rsub = lane >> 3;          // 8 rows per pass, 4 passes
      const float invD = 1.0f / (float)D;
      float gmm[NV][8], ag[NV][8], ab[NV][8];
  #pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(gammap + (v * LPR + lj) * 8), b4 = *reinterpret_cast<const f32x4*>(gammap + (v * LPR + lj) * 8 + 4);
  #pragma unroll
        for (int e = 0; e < 4; ++e) { gmm[v][e] = a4[e]; gmm[v][4 + e] = b4[e]; }
  #pragma unroll
        for (int e = 0; e < 8; ++e) { ag[v][e] = 0.f; ab[v][e] = 0.f; }
      }
  #pragma unroll
      for (int it = 0; it < RW / 8; ++it) {
        const int rl = it * 8 + rsub;
        const long row = row0 + rl;
        const bool ok = row < p.M;
        const long gr = min(row, (long)p.M - 1);
        const float mean = meanp[gr], rstd = rstdp[gr];
        float xh[NV][8], dy[NV][8];
        float s1 = 0.f, s2 = 0.f;
  #pragma unroll
        for (int v = 0; v < NV; ++v) {
          bf16x8 td = *reinterpret_cast<const bf16x8*>(img + rl * LDI + (v * LPR + lj) * 8);
          bf16x8 tz = *reinterpret_cast<const bf16x8*>(Zp + gr * D + (v * LPR + lj) * 8);
          if (!ok) { td = bf16x8{}; tz = bf16x8{}; }
          unpack8(tz, xh[v]);
          unpack8(td, dy[v]);
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[v][e] = ok ? (xh[v][e] - mean) * rstd : 0.f;
            ag[v][e] += dy[v][e] * xh[v][e];
            ab[v][e] += dy[v][e];
            dy[v][e] *= gmm[v][e];
            s1 += dy[v][e];
            s2 += dy[v][e] * xh[v][e];
          }
        }
  #pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const float c1 = s1 * invD, c2 = s2 * invD;
        if (ok) {
  #pragma unroll
          for (int v = 0; v < NV; ++v) {
            float o[8];
  #pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (dy[v][e] - c1 - xh[v][e] * c2);
            const long off = row * D + (v * LPR + lj) * 8;
            bf16x8 ob = pack8(o);
            *reinterpret_cast<bf16x8*>(dZp + off) = ob;
            if (DROP) {
              const uint32_t keep = dropout_keep8(rngx, (uint64_t)off >> 3, thr);
  #pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = ((keep >> e) & 1u) ? o[e] * dsc : 0.f;
              ob = pack8(o);
              *reinterpret_cast<bf16x8*>(dYp + off) = ob;
            }
            if (reload) *reinterpret_cast<bf16x8*>(img + rl * LDI + (v * LPR + lj) * 8) = ob;      // (where this lane read dX1 from)
          }
        }
      }
      if (reload) {                                       // dY rows as activation fragments (the dO fragments are dead)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  #pragma unroll
        for (int rg = 0; rg < RG; ++rg)
  #pragma unroll
          for (int ks = 0; ks < KS1; ++ks)
            xf[rg][ks] = *reinterpret_cast<const bf16x8*>(img + (rg * 16 + c16) * LDI + 32 * ks + 8 * fc_kperm(g));
      }
      // column sums over the wave's 32 rows: lanes with the same lj hold the same columns (8 row slots): fixed-order shuffles
      float* prow = wsp;
  #pragma unroll
      for (int v = 0; v < NV; ++v) {
  #pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = ag[v][e], b2 = ab[v][e];
  #pragma unroll
          for (int o = 8; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b2 += __shfl_xor(b2, o, 64); }
          ag[v][e] = a; ab[v][e] = b2;
        }
        if (rsub == 0) {
          const int c = (v * LPR + lj) * 8;
          *reinterpret_cast<f32x4*>(prow + c) = f32x4{ag[v][0], ag[v][1], ag[v][2], ag[v][3]};
          *reinterpret_cast<f32x4*>(prow + c + 4) = f32x4{ag[v][4], ag[v][5], ag[v][6], ag[v][7]};
          *reinterpret_cast<f32x4*>(prow + D + c) = f32x4{ab[v][0], ab[v][1], ab[v][2], ab[v][3]};
          *reinterpret_cast<f32x4*>(prow + D + c + 4) = f32x4{ab[v][4], ab[v][5], ab[v][6], ab[v][7]};
        }
      }
    }
};
  float* wsum = reinterpret_cast<float*>(smem + FC_NS * SLOT) + wave * (2 * D);      // [2 D]: this wave's column sums, behind the ring
  auto reduce_partial = [&](float* dst) {                // one partial row per WORKGROUP: the waves' sums added in wave order
    for (int i = tid; i < 2 * D; i += NW * 64) {         // (fixed: reproducible) -- seven times fewer rows for the slab reduce
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += reinterpret_cast<const float*>(smem + FC_NS * SLOT)[w * (2 * D) + i];
      dst[(long)blockIdx.x * (2 * D) + i] = t;
    }
  };
  // gate bits of chunk c for this lane's 4 x 8 hidden units: one dword, written by the forward kernel in exactly this wave /
  // chunk / lane order.  Inline-asm load (invisible to the compiler's wait insertion: a tracked register-destination load in
  // the loop would make it drain the ring), issued one chunk ahead, retired by the loop's own counted wait.
  // (address = a wave-uniform base in scalar registers + a 32-bit lane offset, here and for the gH stores below: 64-bit per-lane
  //  pointers live across the chunk loop were what the register allocator spilled -- and reloaded inside the loop)
  const uint32_t* gate_wave = p.gate + (row0 / RW) * nchunk * 64;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  uint32_t gnext;
  auto load_gate = [&](int c) {
    const uint32_t* src = gate_wave + c * 64;           // wave-uniform
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(gnext) : "v"(lane4), "s"(src) : "memory");          // (s_nop: FC_SGPR_HAZARD)
  };
  const uint32_t gh_lane = (uint32_t)((c16 * F + (odd ? 16 + 4 * (g - 1) : 4 * g)) * 2);      // row c16, this lane's 8 units of a pair
  const IqRng rng = DROP ? rng_resolve(p.rng) : p.rng;

  if (PREB) {
    // ---- first stage: acc2 = A0 W0t^T over K0 = 3 D in chunk pairs (slot = two W2-type images: columns 128 i .. +63 | +64 .. +127
    //      of W0t's rows); the last pair of an odd chunk count is half empty ------------------------------------------------------
    constexpr int K0 = 3 * D, NC0 = K0 / FC_CHUNK, NQ0 = (NC0 + 1) / 2;       // 9 chunks in 5 pairs | 6 in 3
    auto issue_pre = [&](int i) {
      unsigned char* slot = smem + (i % FC_NS) * SLOT;
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int pc = min(wave + k * NW, PIECES - 1);
        const int h = pc >= W1_PIECES ? 1 : 0;
        const int l2 = (pc - h * W1_PIECES) * 64 + ln;
        const int r = l2 >> 3, sl = l2 & 7;
        const int q = fc_swz<8>(r, sl);
        const int col = min((2 * i + h) * FC_CHUNK, K0 - FC_CHUNK) + ((q & ~3) | fc_kperm(q & 3)) * 8;
        const unsigned off = (unsigned)((r * K0 + col) * 2);
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(reinterpret_cast<const char*>(p.W0t) + off), (lds_void_t*)(slot + pc * 1024), 16, 0, 0);
      }
    };
    // this wave's A0 rows of pair i as fragments [half][rg][k-step]: inline-asm loads (see load_gate), one pair ahead
    bf16x8 aF[2][2][RG][2];
    auto load_a = [&](int i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (2 * i + h >= NC0) break;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const bf16* src = p.A0 + rowc[rg] * K0 + (2 * i + h) * FC_CHUNK + 32 * jp + 8 * fc_kperm(g);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(aF[i & 1][h][rg][jp]) : "v"(src) : "memory");
          }
      }
    };
    if (have) load_a(0);
    issue_pre(0);
    if (NQ0 > 1) issue_pre(1);
    const IqRng rng0 = DROP ? rng_resolve(p.rng0) : p.rng0;
#pragma unroll
    for (int i = 0; i < NQ0; ++i) {
      // queue, oldest first: pieces of pair i | fragments of pair i | pieces of pair i + 1: at most PPW outstanding = the first two landed
      if (i == 0 || i + 1 >= NQ0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (have && i + 1 < NQ0) load_a(i + 1);
      if (i + 2 < NQ0) issue_pre(i + 2);
      if (!have) continue;
      const uint32_t slot_addr = lds0 + (i % FC_NS) * SLOT;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (2 * i + h >= NC0) break;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) asm volatile("" : "+v"(aF[i & 1][h][rg][jp]));       // (landed: the wait above)
        second_product(slot_addr + h * W1_BYTES - W1_BYTES, aF[i & 1][h]);
      }
    }
    // every wave is done with the ring: the main loop's chunk 0 travels into slot 0 while the images of the tail sit in slots 1, 2
    __syncthreads();
    issue_chunk(0);
    if (have) load_gate(0);
    bf16* img0 = reinterpret_cast<bf16*>(smem + (wave < SLOT / IMG ? SLOT + wave * IMG : 2 * SLOT + (wave - SLOT / IMG) * IMG));
    if (have) ln_tail(p.R0, p.Z0, p.mean0, p.rstd0, p.gamma0, rng0, p.thresh, p.dscale, p.dZ0, p.dY0, img0, wsum, std::true_type{});
    else for (int i = lane; i < 2 * D; i += 64) wsum[i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // chunk 0, the gate word, this stage's stores
    __syncthreads();
    if (nchunk > 1) issue_chunk(1);                      // (the images are dead)
    reduce_partial(p.partial0);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int j = 0; j < NT2; ++j) acc2[rg][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    issue_chunk(0);
    if (nchunk > 1) issue_chunk(1);
    // ---- this wave's dO rows as activation fragments; the gate rows of chunk 0 --------------------------------------------------
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xf[rg][ks] = have ? *reinterpret_cast<const bf16x8*>(p.dO + rowc[rg] * D + 32 * ks + 8 * fc_kperm(g)) : bf16x8{};
    load_gate(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(xf[rg][ks]));
  }
  asm volatile("" : "+v"(gnext));

  for (int c = 0; c < nchunk; ++c) {
    // Queue of this wave at this point, youngest first: the 4 gH stores of chunk c-1 | the gate load of chunk c | the ring
    // pieces of chunk c+1 | ...: "at most 4 outstanding" = gate(c) and ring chunk c+1 (hence c) have landed.  A ragged wave's
    // stores are predicated (their count is not known): it waits for everything.
    if (c > 0) {
      if (full) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * RG) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 2 < nchunk) issue_chunk(c + 2);
    if (!have) continue;
    asm volatile("" : "+v"(gnext));                     // (landed: the wait above; defined from here on as far as the compiler knows)
    const uint32_t gmask = gnext;
    const int f0 = c * FC_CHUNK;
    // ---- first product + gate, 32 hidden units (two column tiles) at a time -- 16 accumulator registers instead of 32: with the
    //      gate rows (16) beside the 144 of the dO fragments and the second product's accumulators, the wide form spilled ------
    const uint32_t slot_addr = lds0 + (c % FC_NS) * SLOT;
    bf16x8 hf[RG][2];
    bf16x8 wp[2][2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x4 acc1[RG][2];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) { acc1[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      auto read_w1 = [&](int ks, bf16x8 (&dst)[2]) {
        const uint32_t a0 = slot_addr + w1at(ks);
        if (jp == 0) { lds_read128<0 * 16 * D * 2>(dst[0], a0); lds_read128<1 * 16 * D * 2>(dst[1], a0); }
        else { lds_read128<2 * 16 * D * 2>(dst[0], a0); lds_read128<3 * 16 * D * 2>(dst[1], a0); }
      };
      read_w1(0, wp[0]);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) { read_w1(ks + 1, wp[(ks + 1) & 1]); FC_LGKM_WAIT(2); }
        else FC_LGKM_WAIT(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int rg = 0; rg < RG; ++rg)
            acc1[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks & 1][t], xf[rg][ks], acc1[rg][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // gate: ReLU and dropout of the FORWARD hidden unit are both "Hid > 0"; the packed tile feeds the second product
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        float w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float va = acc1[rg][0][r], vb = acc1[rg][1][r];
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
          w[r] = __uint_as_float(sw[0]);
          w[4 + r] = __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = ((gmask >> (8 * (2 * jp + rg) + e)) & 1u) ? w[e] * p.gate_scale : 0.f;
        hf[rg][jp] = pack8(w);
      }
    }
    // the next chunk's gate word, ahead of this chunk's stores in the queue
    asm volatile("" ::: "memory");
    if (c + 1 < nchunk) load_gate(c + 1);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const bf16* base = p.gH + (row0 + rg * 16) * F + f0;          // wave-uniform
      if (row0 + rg * 16 + c16 < p.M) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\tglobal_store_dwordx4 %0, %3, %2 offset:64"
                     :: "v"(gh_lane), "v"(hf[rg][0]), "s"(base), "v"(hf[rg][1]) : "memory");          // (s_nop: FC_SGPR_HAZARD; 2 stores)
      }
    }
    // ---- second product: acc2 += gH_c x W1t rows (model columns) 16 j .. +15, hidden units f0 .. f0 + 63 --------------------------
    second_product(slot_addr, hf);
  }
  // ---- tail: norm1 backward on dX1 = acc2 + R.  Image scratch = the ring slots of chunks nchunk-2 and nchunk-3: every wave is past
  //      the last hand-over barrier, nobody reads or fills them any more. ----------------------------------------------------------
  const int sC = (nchunk - 1) % FC_NS;                  // the last chunk's slot (slower waves may still be reading it)
  const int s2 = (sC + 2) % FC_NS, s3 = (sC + 1) % FC_NS;
  bf16* img = reinterpret_cast<bf16*>(smem + (wave < SLOT / IMG ? s2 * SLOT + wave * IMG : s3 * SLOT + (wave - SLOT / IMG) * IMG));
  constexpr int NB0 = D / FC_CHUNK;                     // 64-row blocks of the transposed projection weight: 3 | 2
  auto issue_wot = [&](int b0, int nb, unsigned char* dst) {       // blocks b0 .. b0 + nb - 1 as W1-type images at dst
    const int pieces = nb * W1_PIECES;
    for (int pc = wave; pc < pieces; pc += NW) {
      const int lin = pc * 64 + lane;
      const int r = lin / XCPR, sl = lin - r * XCPR;
      const int q = fc_swz<XCPR>(r & 63, sl);
      const unsigned off = (unsigned)(((b0 * FC_CHUNK + r) * D + ((q & ~3) | fc_kperm(q & 3)) * 8) * 2);
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(reinterpret_cast<const char*>(p.Wot) + off), (lds_void_t*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  if (POSTB) {
    __syncthreads();                                    // every wave has left the loop: the last chunk's slot is free
    issue_wot(0, 2, smem + sC * SLOT);                  // (in flight under the LayerNorm backward below)
  }
  if (have) ln_tail(p.R, p.Z1, p.mean, p.rstd, p.gamma, rng, p.thresh, p.dscale, p.dZ, p.dY, img, wsum, std::integral_constant<bool, POSTB>{});
  else for (int i = lane; i < 2 * D; i += 64) wsum[i] = 0.f;           // a wave without rows contributes zeros
  if (POSTB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of weight blocks 0 and 1 (and its stores)
  __syncthreads();
  if (POSTB && NB0 > 2) issue_wot(2, NB0 - 2, smem + s2 * SLOT);    // (the images are dead now)
  reduce_partial(p.partial);
  if (!POSTB) return;
  // ---- POSTB: dA = dY Wot^T, 64 output columns (one weight block) at a time ----------------------------------------------------
#pragma unroll
  for (int b = 0; b < NB0; ++b) {
    if (b == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (!have) continue;
    const uint32_t blk = lds0 + (b < 2 ? sC * SLOT + b * W1_BYTES : s2 * SLOT);
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x4 acc1[RG][2];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) { acc1[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      bf16x8 wp[2][2];
      auto read_w = [&](int ks, bf16x8 (&dst)[2]) {
        const uint32_t a0 = blk + w1at(ks);
        if (jp == 0) { lds_read128<0 * 16 * D * 2>(dst[0], a0); lds_read128<1 * 16 * D * 2>(dst[1], a0); }
        else { lds_read128<2 * 16 * D * 2>(dst[0], a0); lds_read128<3 * 16 * D * 2>(dst[1], a0); }
      };
      read_w(0, wp[0]);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) { read_w(ks + 1, wp[(ks + 1) & 1]); FC_LGKM_WAIT(2); }
        else FC_LGKM_WAIT(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int rg = 0; rg < RG; ++rg)
            acc1[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks & 1][t], xf[rg][ks], acc1[rg][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int col = FC_CHUNK * b + 32 * jp + (odd ? 16 + 4 * (g - 1) : 4 * g);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        float w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float va = acc1[rg][0][r], vb = acc1[rg][1][r];
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
          w[r] = __uint_as_float(sw[0]);
          w[4 + r] = __uint_as_float(sw[1]);
        }
        const long grow = row0 + rg * 16 + c16;
        if (grow < p.M) *reinterpret_cast<bf16x8*>(p.dA + grow * D + col) = pack8(w);
      }
    }
  }
}

// Rows per wave and waves per workgroup (one workgroup per CU: the ring is most of a CU's LDS).  32-row waves, seven per
// workgroup, cover cfg B's 50,432 rows with 226 workgroups.  Up to 32,768 rows (cfg C at 256 frames: 16,640) that shape would
// leave most CUs idle: 16-row waves, five per workgroup up to 20,480 rows (cfg C: 208 workgroups), eight above.  (Builds with 2
// and 4 waves of 32 rows existed for small M: never faster than the tiled GEMMs there, and -- 24 DMA pieces per wave instead of
// 7 -- they spilled up to 176 scalar and 75 vector registers.)
struct ChainShape { int rg, nw; };
inline ChainShape chain_shape(int M) {
  if (M <= 20480) return ChainShape{1, 5};
  if (M <= 32768) return ChainShape{1, 8};
  return ChainShape{2, FC_MAXW};
}
inline int chain_waves(int M) { return chain_shape(M).nw; }
inline long chain_units(int M) { const int rw = 16 * chain_shape(M).rg; return ((long)M + rw - 1) / rw; }

template <int D, int NW, int RG>
int launch_chain(const FfnChainParams& p, hipStream_t st) {
  constexpr int SLOT = 2 * FC_CHUNK * D * 2;
  const size_t lds = (size_t)FC_NS * SLOT + ((size_t)p.F + 9 * D) * sizeof(float);     // ring (147,456 | 98,304 B) + b1 + nine [D] vectors
  const long units = ((long)p.M + 16 * RG - 1) / (16 * RG);
  const int grid = (int)((units + NW - 1) / NW);
#define FC_LAUNCH(DROP_, MODE_)                                                                                               \
  do {                                                                                                                        \
    auto k = ffn_chain_fwd_kernel<D, NW, RG, DROP_, MODE_>;                                                                   \
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)attr;                                                                                                               \
    k<<<grid, NW * 64, lds, st>>>(p);                                                                                         \
  } while (0)
  if (p.A0 && p.Wq) { if (p.drop1_on) FC_LAUNCH(true, 2); else FC_LAUNCH(false, 2); }
  else if (p.A0) { if (p.drop1_on) FC_LAUNCH(true, 1); else FC_LAUNCH(false, 1); }
  else { if (p.drop1_on) FC_LAUNCH(true, 0); else FC_LAUNCH(false, 0); }
#undef FC_LAUNCH
  return iq_launch_status();
}
template <int D>
int launch_chain_d(const FfnChainParams& p, hipStream_t st) {
  const ChainShape sh = chain_shape(p.M);
  if (sh.rg == 2) return launch_chain<D, FC_MAXW, 2>(p, st);
  return sh.nw == 5 ? launch_chain<D, 5, 1>(p, st) : launch_chain<D, 8, 1>(p, st);
}

template <int D, int NW, int RG>
int launch_chain_bwd(const FfnChainBwdParams& p, hipStream_t st) {
  constexpr int SLOT = 2 * FC_CHUNK * D * 2;
  const size_t lds = (size_t)FC_NS * SLOT + (size_t)NW * 2 * D * sizeof(float);     // ring + the waves' gamma / beta column sums
  const long units = ((long)p.M + 16 * RG - 1) / (16 * RG);
  const int grid = (int)((units + NW - 1) / NW);
#define FC_LAUNCHB(DROP_, MODE_)                                                                                              \
  do {                                                                                                                        \
    auto k = ffn_chain_bwd_kernel<D, NW, RG, DROP_, MODE_>;                                                                   \
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)attr;                                                                                                               \
    k<<<grid, NW * 64, lds, st>>>(p);                                                                                         \
  } while (0)
  if (p.A0) { if (p.drop_on) FC_LAUNCHB(true, 2); else FC_LAUNCHB(false, 2); }
  else if (p.Wot) { if (p.drop_on) FC_LAUNCHB(true, 1); else FC_LAUNCHB(false, 1); }
  else { if (p.drop_on) FC_LAUNCHB(true, 0); else FC_LAUNCHB(false, 0); }
#undef FC_LAUNCHB
  return iq_launch_status();
}
template <int D>
int launch_chain_bwd_d(const FfnChainBwdParams& p, hipStream_t st) {
  const ChainShape sh = chain_shape(p.M);
  if (sh.rg == 2) return launch_chain_bwd<D, FC_MAXW, 2>(p, st);
  return sh.nw == 5 ? launch_chain_bwd<D, 5, 1>(p, st) : launch_chain_bwd<D, 8, 1>(p, st);
}

}  // namespace

// D = 128 | 192, F a multiple of 64 (S: rows per frame -- any; rows are owned by waves, 32 at a time, regardless of frames)
extern "C" int iq_ffn_chain_supported(int S, int D, int F) {
  if (!(D == 128 || D == 192) || F < FC_CHUNK || (F % FC_CHUNK) || S <= 0) return 0;
  return (size_t)FC_NS * 2 * FC_CHUNK * D * 2 + ((size_t)F + 9 * D) * sizeof(float) <= (size_t)160 * 1024 ? 1 : 0;      // F <= 2368 at D = 192
}

namespace {
struct ChainPre {                                        // the optional first stage's operands (iq_attn_out_ffn_chain_fwd)
  const void* A0; const void* W0; const float* b0; const iq_dropout_t* drop0; const void* R0;
  const float* gamma0; const float* beta0; void* Z0; float* mean0; float* rstd0;
  const void* Wq; const float* bq; void* Yq;            // optional last stage (all three or none)
};

int chain_fwd(const ChainPre* pre, const void* X1, const void* W1, const float* b1, const iq_dropout_t* drop1, void* H, const void* W2,
              const float* b2, const iq_dropout_t* drop2, const float* gamma, const float* beta, float eps, void* Z, void* X,
              float* mean, float* rstd, void* gate_bits, int frames, int S, int D, int F, iq_stream_t stream) {
  if (frames <= 0) return IQ_OK;
  if (!X1 || !W1 || !b1 || !H || !W2 || !b2 || !gamma || !beta || !Z || !X || !mean || !rstd) return IQ_ERR_ARG;
  if (!iq_ffn_chain_supported(S, D, F)) return IQ_ERR_UNSUPPORTED;
  if (((uintptr_t)X1 | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)H | (uintptr_t)Z | (uintptr_t)X | (uintptr_t)b1 | (uintptr_t)b2 |
       (uintptr_t)gamma | (uintptr_t)beta) % 16) return IQ_ERR_ARG;
  if ((long)frames * S > 0x7FFFFFFFL) return IQ_ERR_UNSUPPORTED;
  FfnChainParams p = {};
  p.X1 = (const bf16*)X1; p.W1 = (const bf16*)W1; p.W2 = (const bf16*)W2;
  p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.beta = beta;
  p.H = (bf16*)H; p.Z = (bf16*)Z; p.X = (bf16*)X; p.mean = mean; p.rstd = rstd;
  p.gate = (uint32_t*)gate_bits;
  if ((uintptr_t)gate_bits % 16) return IQ_ERR_ARG;
  p.M = frames * S; p.F = F; p.eps = eps;
  auto fill = [](const iq_dropout_t* d, int* on, IqRng* r, uint32_t* th, float* sc) -> bool {
    *on = 0; *th = 0; *sc = 1.f; *r = IqRng{0, 0, 0, nullptr};
    if (d && d->p > 0.f) {
      if (d->p >= 1.f) return false;
      *on = 1;
      r->seed = d->seed; r->step = d->step; r->site = d->site; r->step_dev = d->step_dev;
      *th = dropout_thresh(d->p);
      *sc = dropout_scale(d->p);
    }
    return true;
  };
  if (!fill(drop1, &p.drop1_on, &p.rng1, &p.thresh1, &p.dscale1) || !fill(drop2, &p.drop2_on, &p.rng2, &p.thresh2, &p.dscale2))
    return IQ_ERR_ARG;
  if (pre) {
    if (!pre->A0 || !pre->W0 || !pre->b0 || !pre->R0 || !pre->gamma0 || !pre->beta0 || !pre->Z0 || !pre->mean0 || !pre->rstd0)
      return IQ_ERR_ARG;
    if (((uintptr_t)pre->A0 | (uintptr_t)pre->W0 | (uintptr_t)pre->b0 | (uintptr_t)pre->R0 | (uintptr_t)pre->gamma0 |
         (uintptr_t)pre->beta0 | (uintptr_t)pre->Z0) % 16) return IQ_ERR_ARG;
    p.A0 = (const bf16*)pre->A0; p.W0 = (const bf16*)pre->W0; p.R0 = (const bf16*)pre->R0; p.b0 = pre->b0;
    p.gamma0 = pre->gamma0; p.beta0 = pre->beta0; p.Z0 = (bf16*)pre->Z0; p.X1out = (bf16*)X1; p.mean0 = pre->mean0; p.rstd0 = pre->rstd0;
    if (!fill(pre->drop0, &p.drop0_on, &p.rng0, &p.thresh0, &p.dscale0)) return IQ_ERR_ARG;
    if (pre->Wq || pre->bq || pre->Yq) {
      if (!pre->Wq || !pre->bq || !pre->Yq || ((uintptr_t)pre->Wq | (uintptr_t)pre->bq | (uintptr_t)pre->Yq) % 16) return IQ_ERR_ARG;
      p.Wq = (const bf16*)pre->Wq; p.bq = pre->bq; p.Yq = (bf16*)pre->Yq;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  const double M = (double)p.M;
  double more_bytes = 0.0, more_flops = 0.0;
  if (pre) { more_bytes += 2.0 * (M * D * 3 + (double)D * D) + 8.0 * M; more_flops += 2.0 * M * D * D; }
  if (p.Wq) { more_bytes += 2.0 * (M * 3 * D + 3.0 * D * D); more_flops += 6.0 * M * D * D; }
  IQ_PROF_K(2.0 * (M * D * 3 + M * F + 2.0 * D * F) + 8.0 * M + more_bytes, 4.0 * M * D * F + more_flops,
            "ffn_chain_fwd_kernel<%d, %d, %d, %s, %d>", D, chain_waves(p.M), chain_shape(p.M).rg, p.drop1_on ? "true" : "false", p.Wq ? 2 : pre ? 1 : 0);
  return D == 192 ? launch_chain_d<192>(p, st) : launch_chain_d<128>(p, st);
}
}  // namespace

extern "C" int iq_ffn_chain_fwd(const void* X1, const void* W1, const float* b1, const iq_dropout_t* drop1, void* H,
                                const void* W2, const float* b2, const iq_dropout_t* drop2, const float* gamma,
                                const float* beta, float eps, void* Z, void* X, float* mean, float* rstd, void* gate_bits,
                                int frames, int S, int D, int F, iq_stream_t stream) {
  return chain_fwd(nullptr, X1, W1, b1, drop1, H, W2, b2, drop2, gamma, beta, eps, Z, X, mean, rstd, gate_bits, frames, S, D, F, stream);
}

// The encoder layer from the attention output on, one launch: X1 = norm1(dropout0(A Wo^T + bo) + R) (written, with Z1 / mean1 /
// rstd1, for the backward pass), then iq_ffn_chain_fwd on it; with Wq: Yq[M,3D] = X Wq^T + bq behind it (the next layer's q,k,v).
// Same eps for both norms.
extern "C" int iq_attn_out_ffn_chain_fwd(const void* A, const void* Wo, const float* bo, const iq_dropout_t* drop0, const void* R,
                                         const float* gamma1, const float* beta1, void* Z1, void* X1, float* mean1, float* rstd1,
                                         const void* W1, const float* b1, const iq_dropout_t* drop1, void* H, const void* W2,
                                         const float* b2, const iq_dropout_t* drop2, const float* gamma2, const float* beta2, float eps,
                                         void* Z2, void* X, float* mean2, float* rstd2, void* gate_bits, const void* Wq,
                                         const float* bq, void* Yq, int frames, int S, int D, int F, iq_stream_t stream) {
  const ChainPre pre = {A, Wo, bo, drop0, R, gamma1, beta1, Z1, mean1, rstd1, Wq, bq, Yq};
  return chain_fwd(&pre, X1, W1, b1, drop1, H, W2, b2, drop2, gamma2, beta2, eps, Z2, X, mean2, rstd2, gate_bits, frames, S, D, F, stream);
}

extern "C" int iq_ffn_chain_bwd_partial_rows(int M) {      // one per workgroup
  if (M <= 0) return 0;
  const int nw = chain_waves(M);
  return (int)((chain_units(M) + nw - 1) / nw);
}

extern "C" size_t iq_ffn_chain_gate_bytes(int M, int F) {    // one dword per lane, chunk and wave (a wave = 32 or 16 rows)
  if (M <= 0 || F <= 0) return 0;
  return (size_t)chain_units(M) * (size_t)(F / FC_CHUNK) * 64 * sizeof(uint32_t);
}

namespace {
struct ChainBwdPre {                                     // the optional first stage's operands (iq_qkv_dgrad_ffn_chain_bwd)
  const void* A0; const void* W0t; const void* R0; const void* Z0; const float* mean0; const float* rstd0; const float* gamma0;
  const iq_dropout_t* drop0; void* dZ0; void* dY0; float* partial0;
};

int chain_bwd(const ChainBwdPre* pre, const void* dO, const void* W2t, const void* gate_bits, float gate_scale, void* gH, const void* W1t,
              const void* residual, const void* z1, const float* mean, const float* rstd, const float* gamma, const iq_dropout_t* drop,
              void* dz, void* dy, float* partial, const void* Wot, void* dA, int frames, int S, int D, int F, iq_stream_t stream) {
  if (frames <= 0) return IQ_OK;
  if ((Wot == nullptr) != (dA == nullptr) || ((uintptr_t)Wot | (uintptr_t)dA) % 16) return IQ_ERR_ARG;
  if ((!pre && !dO) || !W2t || !gate_bits || !gH || !W1t || !residual || !z1 || !mean || !rstd || !gamma || !dz || !partial) return IQ_ERR_ARG;
  if (!iq_ffn_chain_supported(S, D, F)) return IQ_ERR_UNSUPPORTED;
  if (((uintptr_t)dO | (uintptr_t)W2t | (uintptr_t)W1t | (uintptr_t)gate_bits | (uintptr_t)gH | (uintptr_t)residual | (uintptr_t)z1 |
       (uintptr_t)dz | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)partial) % 16) return IQ_ERR_ARG;
  if ((long)frames * S > 0x7FFFFFFFL) return IQ_ERR_UNSUPPORTED;
  FfnChainBwdParams p = {};
  p.dO = (const bf16*)dO; p.W2t = (const bf16*)W2t; p.W1t = (const bf16*)W1t;
  p.gate = (const uint32_t*)gate_bits; p.R = (const bf16*)residual; p.Z1 = (const bf16*)z1;
  p.mean = mean; p.rstd = rstd; p.gamma = gamma;
  p.gH = (bf16*)gH; p.dZ = (bf16*)dz; p.dY = (bf16*)dy; p.partial = partial;
  p.M = frames * S; p.F = F; p.gate_scale = gate_scale;
  p.Wot = (const bf16*)Wot; p.dA = (bf16*)dA;
  if (drop && drop->p > 0.f) {
    if (drop->p >= 1.f || !dy) return IQ_ERR_ARG;
    p.drop_on = 1;
    p.rng.seed = drop->seed; p.rng.step = drop->step; p.rng.site = drop->site; p.rng.step_dev = drop->step_dev;
    p.thresh = dropout_thresh(drop->p);
    p.dscale = dropout_scale(drop->p);
  }
  if (pre) {
    if (!Wot) return IQ_ERR_ARG;                        // (built with the last stage only)
    if (!pre->A0 || !pre->W0t || !pre->R0 || !pre->Z0 || !pre->mean0 || !pre->rstd0 || !pre->gamma0 || !pre->dZ0 || !pre->partial0)
      return IQ_ERR_ARG;
    if (((uintptr_t)pre->A0 | (uintptr_t)pre->W0t | (uintptr_t)pre->R0 | (uintptr_t)pre->Z0 | (uintptr_t)pre->gamma0 |
         (uintptr_t)pre->dZ0 | (uintptr_t)pre->dY0 | (uintptr_t)pre->partial0) % 16) return IQ_ERR_ARG;
    const float p0 = pre->drop0 ? pre->drop0->p : 0.f, p1 = drop ? drop->p : 0.f;
    if (p0 != p1 || (p.drop_on && !pre->dY0)) return IQ_ERR_ARG;          // one dropout probability per layer (encoder_layer.py:13-21)
    p.A0 = (const bf16*)pre->A0; p.W0t = (const bf16*)pre->W0t; p.R0 = (const bf16*)pre->R0; p.Z0 = (const bf16*)pre->Z0;
    p.mean0 = pre->mean0; p.rstd0 = pre->rstd0; p.gamma0 = pre->gamma0;
    p.dZ0 = (bf16*)pre->dZ0; p.dY0 = (bf16*)pre->dY0; p.partial0 = pre->partial0;
    if (p.drop_on) { p.rng0.seed = pre->drop0->seed; p.rng0.step = pre->drop0->step; p.rng0.site = pre->drop0->site; p.rng0.step_dev = pre->drop0->step_dev; }
  }
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  const double M = (double)p.M;
  double bytes = 2.0 * (M * D * (4 + (p.drop_on ? 1 : 0) + (Wot ? 1 : 0)) + 2.0 * M * F + 2.0 * D * F + (Wot ? (double)D * D : 0.0)) + 8.0 * M;
  double flops = 4.0 * M * D * F + (Wot ? 2.0 * M * D * D : 0.0);
  if (pre) {      // A0 [M,3D], R0 / Z0 in, dZ0 (+ dY0) out, W0t; dO itself is no longer read
    bytes += 2.0 * (M * 3 * D + M * D * (3 + (p.drop_on ? 1 : 0) - 1) + 3.0 * D * D) + 8.0 * M;
    flops += 6.0 * M * D * D;
  }
  IQ_PROF_K(bytes, flops, "ffn_chain_bwd_kernel<%d, %d, %d, %s, %d>", D, chain_waves(p.M), chain_shape(p.M).rg, p.drop_on ? "true" : "false", pre ? 2 : Wot ? 1 : 0);
  return D == 192 ? launch_chain_bwd_d<192>(p, st) : launch_chain_bwd_d<128>(p, st);
}
}  // namespace

extern "C" int iq_ffn_chain_bwd(const void* dO, const void* W2t, const void* gate_bits, float gate_scale, void* gH, const void* W1t,
                                const void* residual, const void* z1, const float* mean, const float* rstd, const float* gamma,
                                const iq_dropout_t* drop, void* dz, void* dy, float* partial, const void* Wot, void* dA, int frames,
                                int S, int D, int F, iq_stream_t stream) {
  return chain_bwd(nullptr, dO, W2t, gate_bits, gate_scale, gH, W1t, residual, z1, mean, rstd, gamma, drop, dz, dy, partial, Wot, dA,
                   frames, S, D, F, stream);
}

// iq_ffn_chain_bwd with the launch that would produce its dO in front: dX2 = gQKV[M,3D] * Wqkv_t[D,3D]^T + residual0 (the q,k,v
// projection's data gradient of the layer above + the residual path), norm2 backward on it (z2 / mean2 / rstd2 / gamma2, dropout
// site drop2): dz2, dy2 (this layer's dO; dz2 where there is no dropout) and partial2 are written as iq_gemm_bf16_lnbwd writes
// them; `residual` of the second stage is normally dz2 itself (each wave re-reads the rows it wrote).
extern "C" int iq_qkv_dgrad_ffn_chain_bwd(const void* gQKV, const void* Wqkv_t, const void* residual0, const void* z2, const float* mean2,
                                          const float* rstd2, const float* gamma2, const iq_dropout_t* drop2, void* dz2, void* dy2,
                                          float* partial2, const void* W2t, const void* gate_bits, float gate_scale, void* gH,
                                          const void* W1t, const void* residual, const void* z1, const float* mean, const float* rstd,
                                          const float* gamma, const iq_dropout_t* drop, void* dz, void* dy, float* partial, const void* Wot,
                                          void* dA, int frames, int S, int D, int F, iq_stream_t stream) {
  const ChainBwdPre pre = {gQKV, Wqkv_t, residual0, z2, mean2, rstd2, gamma2, drop2, dz2, dy2, partial2};
  return chain_bwd(&pre, nullptr, W2t, gate_bits, gate_scale, gH, W1t, residual, z1, mean, rstd, gamma, drop, dz, dy, partial, Wot, dA,
                   frames, S, D, F, stream);
}
