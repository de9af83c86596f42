// Weight-gradient GEMM  dW[N,K] (+)= dY[M,N]^T * X[M,K],  dbias[N] (+)= colsum(dY)   (gfx950).
//
// This is the autograd backward of every nn.Linear / conv weight on the path (the reference gets
// it from torch.autograd; multi_head_attention.py:11-14, position_wise_feed_forward.py:7-8,
// patch_embedding.py:9).  bf16 operands, fp32 accumulate, fp32 output (gradients stay fp32).
//
// The contraction index m (tokens) is the ROW index of both operands in HBM, so both MFMA
// operands need "8 consecutive m at a fixed column": tiles are staged row-major in LDS exactly as
// they stream from HBM (16 B/lane, rows padded by 32 B) and fragments are fetched with the
// gfx950 transposing read ds_read_b64_tr_b16 (guide T10) -- no software transpose anywhere.
// The bias gradient rides on the same A fragments: one extra MFMA against an all-ones B fragment.
//
// M is large, N*K small: the grid is (output tiles) x (M splits); each workgroup writes an fp32
// partial tile to a slab and wgrad_reduce_kernel sums the slabs in fixed order (bitwise
// reproducible; float atomics would be both slower at this byte rate and order dependent).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include "iqvit.h"
#include "prof.h"

#include "gemm_wgrad_big.h"

namespace {

constexpr int WG_THREADS = 512;   // 8 waves: 2 (n) x 4 (k); two workgroups per CU hide the HBM latency of the single-stage loop
constexpr int TN = 128;     // output rows (n) per tile
constexpr int YLD = TN + 16;  // padded LDS row (elements): 288 B rows -> conflict-free tr reads

struct WgradParams {
  const bf16* Y; const bf16* X;
  int ldy, ldx, M, N, K;
  float* slab;       // [splits][N*K]
  float* bslab;      // [splits][N] or null
  int tiles_n, tiles_k, splits, rows_per_split;
#ifdef IQ_WGRAD_STAMPS
  unsigned long long* stamps;   // diagnostic build only (scripts/wgrad_stamps.py): per-workgroup phase ticks
#endif
};

#ifdef IQ_WGRAD_STAMPS
#define IQ_WTICK(i) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_) :: "memory"); tk_[i] += tn_ - tl_; tl_ = tn_; } while (0)
#else
#define IQ_WTICK(i) do {} while (0)
#endif

__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int ld, int r0, int c0, int lane) {
  // lane group g = lane>>4 takes k-slots {r0+4g+q} U {r0+16+4g+q}, q=0..3; column c0 + (lane&15)
  const int i16 = lane & 15, g = lane >> 4;
  const bf16* a = tile + (r0 + 4 * g + (i16 >> 2)) * ld + c0 + 4 * (i16 & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * ld));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// MC = contraction rows per LDS stage.  One stage is in flight per workgroup while the previous one is multiplied, so
// the bytes in flight per CU (2 workgroups x MC x (TN+TK) x 2 B) against the ~2 us loaded memory latency set the
// ingest rate: MC = 64 measured 7.7 TB/s at the L2 (Little's law limit), MC = 128 doubles the bytes in flight.
template <int TK, int MC>
__global__ __launch_bounds__(WG_THREADS, 4) void wgrad_kernel(const WgradParams p) {
  constexpr int XLD = TK + 16;
  constexpr int KT = TK / 64;       // 16-col k tiles per wave (wave tile = 64 n x TK/4 k)
  constexpr int Y_CH = MC * (TN / 8) / WG_THREADS;  // 2
  constexpr int X_CH = MC * (TK / 8) / WG_THREADS;  // 2 | 1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* Ys = reinterpret_cast<bf16*>(smem);
  bf16* Xs = Ys + MC * YLD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 2, wk = wave & 3;
  const int ntile = p.tiles_n * p.tiles_k;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);   // one split's tiles run on one XCD: slab rows re-read from its L2
  const int split = lid / ntile, tile = lid % ntile;
  const int n0 = (tile / p.tiles_k) * TN, k0 = (tile % p.tiles_k) * TK;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const bool do_bias = p.bslab != nullptr && (tile % p.tiles_k) == 0 && wk == 0;

  f32x4 acc[4][KT], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  bf16x8 ry[Y_CH], rx[X_CH];
  auto gload = [&](int mb) {
#pragma unroll
    for (int c = 0; c < Y_CH; ++c) {
      const int id = tid + c * WG_THREADS, row = id / (TN / 8), ch = id % (TN / 8);
      const int gm = mb + row, gn = n0 + ch * 8;
      bf16x8 v = {};
      if (gm < mend && gn < p.N) v = *reinterpret_cast<const bf16x8*>(p.Y + (long)gm * p.ldy + gn);
      ry[c] = v;
    }
#pragma unroll
    for (int c = 0; c < X_CH; ++c) {
      const int id = tid + c * WG_THREADS, row = id / (TK / 8), ch = id % (TK / 8);
      const int gm = mb + row, gk = k0 + ch * 8;
      bf16x8 v = {};
      if (gm < mend && gk < p.K) v = *reinterpret_cast<const bf16x8*>(p.X + (long)gm * p.ldx + gk);
      rx[c] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int c = 0; c < Y_CH; ++c) {
      const int id = tid + c * WG_THREADS, row = id / (TN / 8), ch = id % (TN / 8);
      *reinterpret_cast<bf16x8*>(Ys + row * YLD + ch * 8) = ry[c];
    }
#pragma unroll
    for (int c = 0; c < X_CH; ++c) {
      const int id = tid + c * WG_THREADS, row = id / (TK / 8), ch = id % (TK / 8);
      *reinterpret_cast<bf16x8*>(Xs + row * XLD + ch * 8) = rx[c];
    }
  };

#ifdef IQ_WGRAD_STAMPS
  unsigned long long tk_[6] = {0, 0, 0, 0, 0, 0}, tl_, tn_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_) :: "memory");
#endif
  if (mbeg < mend) {
    gload(mbeg);
    lstore();
  }
  __syncthreads();
  IQ_WTICK(0);
  for (int mb = mbeg; mb < mend; mb += MC) {
    const bool more = mb + MC < mend;
    if (more) gload(mb + MC);
    IQ_WTICK(1);
#pragma unroll
    for (int s = 0; s < MC / 32; ++s) {
      bf16x8 af[4], bfr[KT];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(Ys, YLD, s * 32, wn * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < KT; ++j) bfr[j] = tr_frag(Xs, XLD, s * 32, wk * (TK / 4) + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
      }
    }
    IQ_WTICK(2);
    __syncthreads();
    IQ_WTICK(3);
    if (more) {
#ifdef IQ_WGRAD_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      IQ_WTICK(4);
#endif
      lstore();
      __syncthreads();
      IQ_WTICK(5);
    }
  }
#ifdef IQ_WGRAD_STAMPS
  if (tid == 0 && p.stamps)
    for (int i = 0; i < 6; ++i) p.stamps[(long)blockIdx.x * 6 + i] = tk_[i];
#endif

  float* out = p.slab + (long)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (n < p.N) {
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int k = k0 + wk * (TK / 4) + j * 16 + (lane & 15);
          if (k < p.K) out[(long)n * p.K + k] = acc[i][j][r];
        }
        if (do_bias && (lane & 15) == 0) p.bslab[(long)split * p.N + n] = accb[i][r];
      }
    }
  }
}

// ---- wave-private variant (small N x K, the ViT-Tiny / raw-IQ shapes) ------------------------------------------------
// Phase stamps of the kernel above on cfg B (scripts/wgrad_stamps.py) showed the data always there in time (1-3 % of a
// workgroup's life waiting for HBM) and the life going to the two workgroup barriers per stage (37 %), the burst of
// loads all 16 waves of a CU issue at the same moment (20 %) and LDS fragment reads (35 %: a 64x16 / 64x32 wave tile
// re-reads 0.75-1.25 fragments per MFMA).  Here nothing is shared between waves, so there is nothing to synchronise:
// every wave owns the WHOLE 64x64 output tile for its own 32-row stages of the split (stage s -> wave s mod 4), stages
// its rows through a private LDS area only to transpose them (ds_read_b64_tr_b16), 0.5 fragments per MFMA, and the
// waves free-run; the four accumulator sets meet once, through LDS, when the split is done.
// Round 2: the wave's tile is 64 x 64, 128 x 64 or 64 x 128 (n x k), chosen per problem along its wider side.  The
// kernel moves its operands at the L2-level rate the chip gives this access shape (~14.5 TB/s: 8 waves x 16 KB in
// flight measured the same as 12 x 8 KB), so the lever is bytes per flop: a 64 x 64 tile loads 256 B per contraction row
// for 8192 flop, a 128 x 64 tile 384 B for 16384 (0.75 x), and the layer's launch moves 1.10 GB instead of 1.39 GB.
constexpr int PW_THREADS = 256, PW_T = 64, PW_ROWS = 32;
constexpr int PW_MAXW = 128;                        // widest tile side
constexpr int pw_ld(int t) { return t + 16; }       // padded LDS row (elements): conflict-free transposing reads
constexpr int PW_WAVE_LDS = PW_ROWS * (pw_ld(PW_MAXW) + pw_ld(PW_T));   // elements per wave: widest Y stage + X stage
constexpr int PW_RED_ROWS = 16;                     // n-rows of the final cross-wave sum per pass

// A launch covers up to PW_MAXP problems that share M (the four Linear layers of one encoder layer): the grid is
// (all their tiles) x (M splits), so one pipeline fill / drain and one slab reduce serve the whole layer.
constexpr int PW_MAXP = 4;
struct PwProb {
  const bf16* Y; const bf16* X;
  float* slab;       // [splits][N*K]
  float* bslab;      // [splits][N] or null
  int ldy, ldx, N, K, tiles_n, tiles_k, tile0;
  int tn, tk;        // the wave tile of this problem: 64 x 64, 128 x 64 or 64 x 128
};
struct PwGroup {
  PwProb pr[PW_MAXP];
  int nprob, M, ntile, splits, rows_per_split;
#ifdef IQ_WGRAD_STAMPS
  unsigned long long* stamps;
#endif
};

template <int TN, int TK>
__device__ __forceinline__ void pw_body(const PwGroup& g, const PwProb& p, int tile, int split, unsigned char* smem) {
  constexpr int NI = TN / 16, NJ = TK / 16;            // 16 x 16 accumulator tiles
  constexpr int LDY = pw_ld(TN), LDX = pw_ld(TK);
  constexpr int YC = TN / 64, XC = TK / 64;            // 64-column load groups per row
  constexpr int RED_LD = TK + 4;                       // padded fp32 row of the final cross-wave sum
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16* Ys = reinterpret_cast<bf16*>(smem) + wave * PW_WAVE_LDS;
  bf16* Xs = Ys + PW_ROWS * LDY;
  // Tiles run along the LONGER side of the output first: where consecutive workgroups spill over to the next XCD the
  // cut then separates blocks of the wide operand, and only the narrow one is fetched by both L2s.
  int nt, kt;
  if (p.tiles_n >= p.tiles_k) { nt = tile / p.tiles_k; kt = tile % p.tiles_k; }
  else { kt = tile / p.tiles_n; nt = tile % p.tiles_n; }
  const int n0 = nt * TN, k0 = kt * TK;
  const int mbeg = split * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);
  const bool do_bias = p.bslab != nullptr && kt == 0;

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dY: accumulated by the VALU from the staged rows, 8 columns per lane and 64-column
  // group (round 1 spent one extra MFMA accumulator tile per 16 columns on it: 32 registers at TN = 128, which spilled)
  float bs[YC][8];
#pragma unroll
  for (int c = 0; c < YC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[c][e] = 0.f;

  // a load instruction covers 8 rows x 128 B; the 16-lane groups of the LDS write pair rows r and r+4, whose
  // padded images fall on disjoint bank halves
  const int prow = ((lane >> 4) & 3) + ((lane >> 3) & 1) * 4, pch = lane & 7;
  // columns past N / K only feed accumulator rows / columns that are never stored: point them at column 0 instead
  // of predicating the loads.  Rows past the split end must contribute zero (only a split's last stage can be partial).
  const bf16* yptr[YC];
  const bf16* xptr[XC];
#pragma unroll
  for (int c = 0; c < YC; ++c) { const int col = n0 + c * 64 + pch * 8; yptr[c] = p.Y + (long)prow * p.ldy + (col < p.N ? col : 0); }
#pragma unroll
  for (int c = 0; c < XC; ++c) { const int col = k0 + c * 64 + pch * 8; xptr[c] = p.X + (long)prow * p.ldx + (col < p.K ? col : 0); }
  const long ystep = 8L * p.ldy, xstep = 8L * p.ldx;
  bf16x8 ry[YC][4], rx[XC][4];
  auto gload = [&](int m0) {
    const bool full = m0 + PW_ROWS <= mend;           // wave-uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = full || (m0 + j * 8 + prow < mend);
#pragma unroll
      for (int c = 0; c < YC; ++c) {
        bf16x8 v = {};
        if (ok) v = *reinterpret_cast<const bf16x8*>(yptr[c] + (long)m0 * p.ldy + j * ystep);
        ry[c][j] = v;
      }
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        bf16x8 v = {};
        if (ok) v = *reinterpret_cast<const bf16x8*>(xptr[c] + (long)m0 * p.ldx + j * xstep);
        rx[c][j] = v;
      }
    }
  };
  auto lstore = [&]() {
    if (do_bias) {                                    // workgroup-uniform
#pragma unroll
      for (int c = 0; c < YC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bs[c][e] += (float)ry[c][j][e];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < YC; ++c) *reinterpret_cast<bf16x8*>(Ys + (j * 8 + prow) * LDY + c * 64 + pch * 8) = ry[c][j];
#pragma unroll
      for (int c = 0; c < XC; ++c) *reinterpret_cast<bf16x8*>(Xs + (j * 8 + prow) * LDX + c * 64 + pch * 8) = rx[c][j];
    }
  };

  int m = mbeg + wave * PW_ROWS, mnext = m + 4 * PW_ROWS;
  if (m < mend) {
    gload(m);
    lstore();
    if (mnext < mend) gload(mnext);
  }
  while (m < mend) {
    // LDS is in-order within a wave; only keep the compiler from moving reads across the writes of other lanes
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    bf16x8 af[NI], bfr[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) af[i] = tr_frag(Ys, LDY, 0, i * 16, lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bfr[j] = tr_frag(Xs, LDX, 0, j * 16, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    m = mnext; mnext += 4 * PW_ROWS;
    // the stage lives in registers: its LDS image is overwritten (and the stage after it requested) while the MFMAs
    // above drain; a wave that finds its data late only stalls itself
    if (m < mend) {
      lstore();
      if (mnext < mend) gload(mnext);
    }
  }

  // ---- sum the four waves' accumulators (PW_RED_ROWS n-rows at a time through LDS), write the slab ----------------
  float* red = reinterpret_cast<float*>(smem);     // [4 waves][PW_RED_ROWS][RED_LD]
  float* out = p.slab + (long)split * p.N * p.K;
  static_assert(4 * PW_RED_ROWS * RED_LD * 4 <= 4 * PW_WAVE_LDS * 2, "the reduction area fits the stage areas");
#pragma unroll
  for (int h = 0; h < NI; ++h) {                   // one 16-row accumulator tile row per pass
    __syncthreads();                               // stage areas (h = 0) / previous pass no longer read
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(wave * PW_RED_ROWS + (lane >> 4) * 4 + r) * RED_LD + j * 16 + (lane & 15)] = acc[h][j][r];
    __syncthreads();
    constexpr int F4 = PW_RED_ROWS * TK / 4;       // float4 per pass
#pragma unroll
    for (int c = 0; c < (F4 + PW_THREADS - 1) / PW_THREADS; ++c) {
      const int id = tid + c * PW_THREADS;
      if (id < F4) {
        const int row = id / (TK / 4), c4 = (id % (TK / 4)) * 4;
        f32x4 sum = *reinterpret_cast<const f32x4*>(red + row * RED_LD + c4);
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const f32x4*>(red + (w * PW_RED_ROWS + row) * RED_LD + c4);
        const int n = n0 + h * 16 + row, k = k0 + c4;
        if (n < p.N && k < p.K) *reinterpret_cast<f32x4*>(out + (long)n * p.K + k) = sum;   // K % 8 == 0
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    // a lane's 8 columns were summed over its own rows (prow + 8 j of every stage): add the 8 lanes that share pch
    // (lane bits 3, 4, 5), fixed order; the pch leaders publish
#pragma unroll
    for (int c = 0; c < YC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bs[c][e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8) red[wave * TN + c * 64 + pch * 8 + e] = v;
      }
    __syncthreads();
    if (tid < TN && n0 + tid < p.N)
      p.bslab[(long)split * p.N + n0 + tid] = red[tid] + red[TN + tid] + red[2 * TN + tid] + red[3 * TN + tid];
  }
}

__global__ __launch_bounds__(PW_THREADS, 2) void wgrad_pw_kernel(const PwGroup g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);   // the tiles of one split are neighbours on one XCD: re-reads hit its L2
  const int split = lid / g.ntile, gt = lid % g.ntile;
  PwProb p = g.pr[0];
#pragma unroll
  for (int i = 1; i < PW_MAXP; ++i)
    if (i < g.nprob && gt >= g.pr[i].tile0) p = g.pr[i];
  const int tile = gt - p.tile0;
  if (p.tn == 128) pw_body<128, 64>(g, p, tile, split, smem);          // wave-uniform (a workgroup serves one problem)
  else if (p.tk == 128) pw_body<64, 128>(g, p, tile, split, smem);
  else pw_body<64, 64>(g, p, tile, split, smem);
}

// out[i] (+)= sum_s slab[s][i] for up to 2*PW_MAXP segments (weights and biases of a group) in ONE launch.
// Block = 64 float4 columns x 4 split slices (coalesced 1 KiB rows, 4x the loads in flight of a one-thread-per-column
// loop), slices combined through LDS in fixed order: bit-reproducible.
constexpr int RED_MAXX = 4;   // extra (LayerNorm partial) segments
struct RedSeg { const float* slab; float* out; long n, stride; int rows, blk0, tall; };
struct RedGroup { RedSeg s[2 * PW_MAXP + RED_MAXX]; int nseg, accumulate; };

// Block shape per segment: "wide" = 64 float4 columns x 4 row slices (few slab rows, many columns), "tall" = 16 columns
// x 16 slices (hundreds of LayerNorm partial rows, 192 columns): keeps every thread's serial chain short either way.
inline int red_blocks(long n, int tall) { return (int)((n + (tall ? 63 : 255)) / (tall ? 64 : 256)); }

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedGroup g) {
  __shared__ f32x4 part[256];
  RedSeg sg = g.s[0];
#pragma unroll
  for (int i = 1; i < 2 * PW_MAXP + RED_MAXX; ++i)
    if (i < g.nseg && (int)blockIdx.x >= g.s[i].blk0) sg = g.s[i];
  const float* __restrict__ slab = sg.slab;
  float* __restrict__ out = sg.out;
  const long n = sg.n, stride = sg.stride;
  const long blk = (long)blockIdx.x - sg.blk0;
  const int rows = sg.rows, accumulate = g.accumulate;
  const int ncol = sg.tall ? 16 : 64, nsl = 256 / ncol;
  const int col = threadIdx.x % ncol, sl = threadIdx.x / ncol;
  const long i = (blk * ncol + col) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i + 4 <= n) {
#pragma unroll 8
    for (int sp = sl; sp < rows; sp += nsl) s += *reinterpret_cast<const f32x4*>(slab + (long)sp * stride + i);
  } else if (i < n) {
    for (int sp = sl; sp < rows; sp += nsl)
      for (int e = 0; e < 4 && i + e < n; ++e) s[e] += slab[(long)sp * stride + i + e];
  }
  part[sl * ncol + col] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    f32x4 t = part[col];
    for (int k = 1; k < nsl; ++k) t += part[k * ncol + col];
    if (i + 4 <= n) {
      f32x4* o = reinterpret_cast<f32x4*>(out + i);
      *o = accumulate ? *o + t : t;
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = accumulate ? out[i + e] + t[e] : t[e];
    }
  }
}

struct WgradPlan { int tk, mc, tiles_n, tiles_k, splits, rows_per_split; };

// shared-tile kernel (one problem)
inline WgradPlan wgrad_plan(int M, int N, int K) {
  WgradPlan w;
  w.tk = (K % 128 == 0 || K > 512) ? 128 : 64;
  w.mc = 64;      // 128-row stages measured equal on cfg B and spill 5 VGPRs in the 128x128 variant
  const int MC = w.mc;
  w.tiles_n = (N + TN - 1) / TN;
  w.tiles_k = (K + w.tk - 1) / w.tk;
  const int tiles = w.tiles_n * w.tiles_k;
  // Two 8-wave workgroups fit a CU: fill the 512 slots ONCE (one workgroup more costs a whole extra round).
  int splits = 512 / tiles;
  const int max_splits = (M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (M + splits - 1) / splits;
  rps = ((rps + MC - 1) / MC) * MC;
  w.rows_per_split = rps;
  w.splits = (M + rps - 1) / rps;
  return w;
}

// Wave-private 64x64 tiles while the outputs are small (operand re-reads grow as N*K/32 per row and stay in L2);
// the shared 128x128 tiles for the large, MFMA-bound shapes (ViT-Base).
inline bool pw_eligible(int N, int K) { return (long)N * K <= 512 * 1024; }
inline size_t pad4(size_t v) { return (v + 3) / 4 * 4; }

// wave tile of a problem: 128 along the wider side when that side holds at least two such tiles, else 64 x 64
inline void pw_tile(int N, int K, int* tn, int* tk) {
  *tn = 64; *tk = 64;
  if (N >= K && N >= 256) *tn = 128;
  else if (K > N && K >= 256) *tk = 128;
}
struct PwPlan { int ntile, splits, rows_per_split; size_t floats; };
inline PwPlan pw_plan(const iq_wgrad_problem_t* pr, int nprob, int M, int max_wgs) {
  PwPlan w;
  w.ntile = 0;
  for (int i = 0; i < nprob; ++i) {
    int tn, tk;
    pw_tile(pr[i].N, pr[i].K, &tn, &tk);
    w.ntile += ((pr[i].N + tn - 1) / tn) * ((pr[i].K + tk - 1) / tk);
  }
  // two 4-wave workgroups per CU (a 128 x 64 wave tile holds 128 accumulator registers), filled once -- or the caller's
  // smaller budget when the launch is meant to share the CUs with another stream's kernels
  const int slots = (max_wgs > 0 && max_wgs < 512) ? max_wgs : 512;
  int splits = slots / w.ntile;
  const int max_splits = (M + 255) / 256;           // at least two stages per wave
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int unit = 4 * PW_ROWS;
  int rps = (M + splits - 1) / splits;
  rps = ((rps + unit - 1) / unit) * unit;
  w.rows_per_split = rps;
  w.splits = (M + rps - 1) / rps;
  w.floats = 0;
  for (int i = 0; i < nprob; ++i) w.floats += (size_t)w.splits * (pad4((size_t)pr[i].N * pr[i].K) + pad4(pr[i].N));
  return w;
}
inline bool group_is_pw(const iq_wgrad_problem_t* pr, int nprob) {
  if (nprob < 1 || nprob > PW_MAXP) return false;
  for (int i = 0; i < nprob; ++i)
    if (!pw_eligible(pr[i].N, pr[i].K)) return false;
  return true;
}
inline size_t shared_ws_floats(int M, int N, int K) {
  const WgradPlan w = wgrad_plan(M, N, K);
  return (size_t)w.splits * (pad4((size_t)N * K) + pad4(N));
}

#ifdef IQ_WGRAD_STAMPS
unsigned long long* g_wstamps = nullptr;
#endif

int launch_reduce(const RedGroup& rg, int nblk, hipStream_t st) {
  IQ_PROF(IQ_FAM_WGRAD, st);
  wgrad_reduce_kernel<<<nblk, 256, 0, st>>>(rg);
  // (no algorithmic bytes of its own: the gradient it writes is counted with the partial-tile kernel; its slab traffic exists
  //  because of the M-split and shows up as this launch's time)
  IQ_PROF_K(0.0, 0.0, "wgrad_reduce_kernel");
  return IQ_OK;
}

// algorithmic work of a weight-gradient group: both operands read once, fp32 results written once (slabs excluded)
void wgrad_work(const iq_wgrad_problem_t* pr, int nprob, int M, double* bytes, double* flops) {
  *bytes = *flops = 0;
  for (int i = 0; i < nprob; ++i) {
    *bytes += 2.0 * M * ((double)pr[i].N + pr[i].K) + 4.0 * (double)pr[i].N * pr[i].K;
    *flops += 2.0 * M * (double)pr[i].N * pr[i].K;
  }
}

int wgrad_shared_one(const iq_wgrad_problem_t& pb, int M, float* ws, int accumulate, hipStream_t st) {
  const int N = pb.N, K = pb.K;
  const WgradPlan w = wgrad_plan(M, N, K);
  WgradParams q;
  q.Y = (const bf16*)pb.dY; q.X = (const bf16*)pb.X; q.ldy = pb.ldy; q.ldx = pb.ldx; q.M = M; q.N = N; q.K = K;
  q.slab = ws;
  q.bslab = pb.dbias ? ws + (size_t)w.splits * pad4((size_t)N * K) : nullptr;
  q.tiles_n = w.tiles_n; q.tiles_k = w.tiles_k; q.splits = w.splits; q.rows_per_split = w.rows_per_split;
#ifdef IQ_WGRAD_STAMPS
  q.stamps = g_wstamps;
#endif
  const int grid = w.tiles_n * w.tiles_k * w.splits;
  const size_t lds = (size_t)w.mc * (YLD + w.tk + 16) * 2;
  {
  IQ_PROF(IQ_FAM_WGRAD, st);
#define IQ_WG_LAUNCH(TK_, MC_)                                                                                   \
  do {                                                                                                           \
    auto k = wgrad_kernel<TK_, MC_>;                                                                             \
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    k<<<grid, WG_THREADS, lds, st>>>(q);                                                                         \
  } while (0)
  if (w.tk == 128) IQ_WG_LAUNCH(128, 64);
  else IQ_WG_LAUNCH(64, 64);
#undef IQ_WG_LAUNCH
  double wbytes, wflops;
  wgrad_work(&pb, 1, M, &wbytes, &wflops);
  IQ_PROF_K(wbytes, wflops, "wgrad_kernel<%d, 64>", w.tk == 128 ? 128 : 64);
  }
  RedGroup rg;
  memset(&rg, 0, sizeof(rg));
  const long n = (long)N * K;
  rg.s[0] = RedSeg{q.slab, pb.dW, n, n, w.splits, 0, 0};
  int nblk = (int)((n + 255) / 256);
  rg.nseg = 1;
  if (pb.dbias) { rg.s[1] = RedSeg{q.bslab, pb.dbias, (long)N, (long)N, w.splits, nblk, 0}; nblk += (N + 255) / 256; rg.nseg = 2; }
  rg.accumulate = accumulate;
  return launch_reduce(rg, nblk, st);
}

}  // namespace

#ifdef IQ_WGRAD_STAMPS
extern "C" void iq_debug_set_wgrad_stamps(unsigned long long* p) { g_wstamps = p; }
#endif

extern "C" size_t iq_wgrad_grouped_ws_bytes(const iq_wgrad_problem_t* probs, int nprob, int M, int max_workgroups) {
  if (!probs || nprob <= 0 || M <= 0) return 0;
  // (sized for every kernel the call may choose: the choice also depends on operand alignment, unknown here)
  WbPlan wb;
  iq_wgrad_problem_t al[PW_MAXP];
  size_t big = 0;
  if (nprob <= PW_MAXP) {
    for (int i = 0; i < nprob; ++i) { al[i] = probs[i]; al[i].dY = nullptr; al[i].X = nullptr; al[i].ldy = 64; al[i].ldx = 64; }
    if (wgrad_big_plan(al, nprob, M, &wb)) big = wb.floats * sizeof(float);
  }
  if (group_is_pw(probs, nprob)) {
    const size_t b = pw_plan(probs, nprob, M, max_workgroups).floats * sizeof(float);
    return b > big ? b : big;
  }
  size_t mx = big / sizeof(float);                  // run one at a time, sharing the area
  for (int i = 0; i < nprob; ++i) {
    const size_t b = pw_eligible(probs[i].N, probs[i].K) ? pw_plan(probs + i, 1, M, max_workgroups).floats
                                                          : shared_ws_floats(M, probs[i].N, probs[i].K);
    if (b > mx) mx = b;
  }
  return mx * sizeof(float);
}

extern "C" int iq_gemm_bf16_wgrad_grouped(const iq_wgrad_problem_t* probs, int nprob, int M, float* ws, size_t ws_bytes,
                                          int accumulate, int max_workgroups, const iq_reduce_seg_t* extra, int nextra,
                                          iq_stream_t stream) {
  if (nextra < 0 || nextra > RED_MAXX || (nextra > 0 && !extra)) return IQ_ERR_ARG;
  for (int i = 0; i < nextra; ++i)
    if (!extra[i].partials || !extra[i].out || extra[i].rows <= 0 || extra[i].n <= 0 || extra[i].row_stride < extra[i].n ||
        (extra[i].row_stride % 4) || (((uintptr_t)extra[i].partials | (uintptr_t)extra[i].out) & 15))
      return IQ_ERR_ARG;
  auto add_extra = [&](RedGroup& rg, int& nblk) {
    for (int i = 0; i < nextra; ++i) {
      const int tall = extra[i].rows >= 64;
      rg.s[rg.nseg++] = RedSeg{extra[i].partials, extra[i].out, (long)extra[i].n, (long)extra[i].row_stride, extra[i].rows, nblk, tall};
      nblk += red_blocks(extra[i].n, tall);
    }
  };
  if (nprob <= 0 && nextra == 0) return IQ_OK;
  if (nprob > 0 && (!probs || !ws || M <= 0)) return IQ_ERR_ARG;
  for (int i = 0; i < nprob; ++i) {
    const iq_wgrad_problem_t& b = probs[i];
    if (b.N <= 0 || b.K <= 0 || !b.dY || !b.X || !b.dW) return IQ_ERR_ARG;
    if ((b.N % 8) || (b.K % 8) || (b.ldy % 8) || (b.ldx % 8)) return IQ_ERR_UNSUPPORTED;
    if (((uintptr_t)b.dW & 15) != 0) return IQ_ERR_ARG;
  }
  if (ws_bytes < iq_wgrad_grouped_ws_bytes(probs, nprob, M, max_workgroups)) return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  double wbytes, wflops;
  wgrad_work(probs, nprob, M, &wbytes, &wflops);
  WbPlan wb;
  if (nprob > 0 && max_workgroups == 0 && wgrad_big_plan(probs, nprob, M, &wb) && ws_bytes >= wb.floats * sizeof(float)) {
    // LDS-shared 256-row tiles (gemm_wgrad_big.hip): whole 64-row steps, line-aligned operands, one common column tile
    float *slab[PW_MAXP], *bslab[PW_MAXP];
    {
      IQ_PROF(IQ_FAM_WGRAD, st);
      wgrad_big_launch(probs, nprob, M, wb, ws, slab, bslab, st);
      IQ_PROF_K(wbytes, wflops, "wgrad_big_kernel<%d>", wb.tk);
    }
    RedGroup rg;
    memset(&rg, 0, sizeof(rg));
    int nblk = 0;
    for (int i = 0; i < nprob; ++i) {
      const long n = (long)probs[i].N * probs[i].K;
      rg.s[rg.nseg++] = RedSeg{slab[i], probs[i].dW, n, (long)pad4((size_t)n), wb.splits, nblk, 0};
      nblk += (int)((n + 255) / 256);
      if (probs[i].dbias) {
        rg.s[rg.nseg++] = RedSeg{bslab[i], probs[i].dbias, (long)probs[i].N, (long)pad4(probs[i].N), wb.splits, nblk, 0};
        nblk += (probs[i].N + 255) / 256;
      }
    }
    add_extra(rg, nblk);
    rg.accumulate = accumulate;
    launch_reduce(rg, nblk, st);
    return iq_launch_status();
  }
  if (!group_is_pw(probs, nprob)) {
    for (int i = 0; i < nprob; ++i) {
      int rc;
      if (pw_eligible(probs[i].N, probs[i].K)) {
        rc = iq_gemm_bf16_wgrad_grouped(probs + i, 1, M, ws, ws_bytes, accumulate, max_workgroups, nullptr, 0, stream);
      } else {
        rc = wgrad_shared_one(probs[i], M, ws, accumulate, st);
      }
      if (rc != IQ_OK) return rc;
    }
    if (nextra > 0) {
      RedGroup rg;
      memset(&rg, 0, sizeof(rg));
      int nblk = 0;
      add_extra(rg, nblk);
      rg.accumulate = accumulate;
      launch_reduce(rg, nblk, st);
    }
    return iq_launch_status();
  }
  const PwPlan w = pw_plan(probs, nprob, M, max_workgroups);
  PwGroup g;
  RedGroup rg;
  memset(&g, 0, sizeof(g));
  memset(&rg, 0, sizeof(rg));
  g.nprob = nprob; g.M = M; g.ntile = w.ntile; g.splits = w.splits; g.rows_per_split = w.rows_per_split;
#ifdef IQ_WGRAD_STAMPS
  g.stamps = g_wstamps;
#endif
  float* cur = ws;
  int tile0 = 0, nblk = 0;
  for (int i = 0; i < nprob; ++i) {
    const iq_wgrad_problem_t& b = probs[i];
    PwProb& q = g.pr[i];
    q.Y = (const bf16*)b.dY; q.X = (const bf16*)b.X; q.ldy = b.ldy; q.ldx = b.ldx; q.N = b.N; q.K = b.K;
    pw_tile(b.N, b.K, &q.tn, &q.tk);
    q.tiles_n = (b.N + q.tn - 1) / q.tn;
    q.tiles_k = (b.K + q.tk - 1) / q.tk;
    q.tile0 = tile0;
    tile0 += q.tiles_n * q.tiles_k;
    const long n = (long)b.N * b.K;
    q.slab = cur; cur += (size_t)w.splits * pad4((size_t)n);
    rg.s[rg.nseg++] = RedSeg{q.slab, b.dW, n, n, w.splits, nblk, 0};
    nblk += (int)((n + 255) / 256);
    if (b.dbias) {
      q.bslab = cur; cur += (size_t)w.splits * pad4(b.N);
      rg.s[rg.nseg++] = RedSeg{q.bslab, b.dbias, (long)b.N, (long)b.N, w.splits, nblk, 0};
      nblk += (b.N + 255) / 256;
    }
  }
  add_extra(rg, nblk);
  rg.accumulate = accumulate;
  const size_t lds_pw = (size_t)4 * PW_WAVE_LDS * 2;   // 4 waves x 32 rows x (144 + 80) elements = 56 KiB
  if (lds_pw > 48 * 1024) (void)hipFuncSetAttribute((const void*)wgrad_pw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pw);
  {
    IQ_PROF(IQ_FAM_WGRAD, st);
    wgrad_pw_kernel<<<w.ntile * w.splits, PW_THREADS, lds_pw, st>>>(g);
    IQ_PROF_K(wbytes, wflops, "wgrad_pw_kernel");
  }
  launch_reduce(rg, nblk, st);
  return iq_launch_status();
}

extern "C" size_t iq_wgrad_ws_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  iq_wgrad_problem_t b;
  memset(&b, 0, sizeof(b));
  b.N = N; b.K = K;
  return iq_wgrad_grouped_ws_bytes(&b, 1, M, 0);
}

extern "C" int iq_gemm_bf16_wgrad(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dbias, int M,
                                  int N, int K, float* ws, size_t ws_bytes, int accumulate, iq_stream_t stream) {
  if (N <= 0 || K <= 0) return IQ_OK;
  iq_wgrad_problem_t b;
  b.dY = dY; b.ldy = ldy; b.X = X; b.ldx = ldx; b.dW = dW; b.dbias = dbias; b.N = N; b.K = K;
  return iq_gemm_bf16_wgrad_grouped(&b, 1, M, ws, ws_bytes, accumulate, 0, nullptr, 0, stream);
}
