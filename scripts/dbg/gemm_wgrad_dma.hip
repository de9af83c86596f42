// Weight-gradient GEMM  dW[N,K] (+)= dY[M,N]^T * X[M,K],  dbias[N] (+)= colsum(dY)   (gfx950).
//
// This is the autograd backward of every nn.Linear / conv weight on the path (the reference gets
// it from torch.autograd; multi_head_attention.py:11-14, position_wise_feed_forward.py:7-8,
// patch_embedding.py:9).  bf16 operands, fp32 accumulate, fp32 output (gradients stay fp32).
//
// The contraction index m (tokens) is the ROW index of both operands in HBM, so both MFMA
// operands need "8 consecutive m at a fixed column": tiles are staged row-major in LDS exactly as
// they stream from HBM and fragments are fetched with the gfx950 transposing read
// ds_read_b64_tr_b16 (guide T10) -- no software transpose anywhere.  Tiles travel HBM -> LDS by
// global_load_lds_dwordx4 through a 4-slot ring (three 32-row stages in flight, counted vmcnt + raw
// s_barrier); the register-staged predecessor of this loop kept one stage in flight and ran at the
// HBM latency per 64 rows (1.6 TB/s algorithmic).  Bank conflicts are avoided by an XOR swizzle applied to
// the DMA source address and to the transposing reads (unpadded rows: the DMA writes LDS lane-linearly).
// The bias gradient rides on the same A fragments: one extra MFMA against an all-ones B fragment.
//
// M is large, N*K small: the grid is (output tiles) x (M splits); each workgroup writes an fp32
// partial tile to a slab and wgrad_reduce_kernel sums the slabs in fixed order (bitwise
// reproducible; float atomics would be both slower at this byte rate and order dependent).
#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int WG_THREADS = 512;   // 8 waves: 2 (n) x 4 (k)
constexpr int TN = 128;           // output rows (n) per tile
constexpr int MC = 32;            // contraction rows per stage (one MFMA k-step)

struct WgradParams {
  const bf16* Y; const bf16* X;
  int ldy, ldx, M, N, K;
  float* slab;       // [splits][N*K]
  float* bslab;      // [splits][N] or null
  int tiles_n, tiles_k, splits, rows_per_split;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// Row-major LDS tile with ROWB-byte rows (256 or 128) and NO padding (the DMA writes LDS lane-linearly);
// 32-byte chunk c of row r lives at chunk c ^ f(r), f = r&7 (256 B rows) or (r>>1)&3 (128 B rows), so the
// 8 rows x 32 B a half-wave touches in one transposing read fall on 8 distinct bank groups.
template <int ROWB> __device__ __forceinline__ int swz32(int row) { return ROWB == 256 ? (row & 7) : ((row >> 1) & 3); }

// lane group g takes k-slots {r0+4g+q} U {r0+16+4g+q}, q=0..3; column c0 + (lane&15)   (c0 % 16 == 0)
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int r0, int c0, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * g + (i16 >> 2);
  const unsigned char* a = tile + row * ROWB + (((c0 >> 4) ^ swz32<ROWB>(row)) << 5) + ((i16 & 3) << 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * ROWB));   // f(row+16) == f(row)
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int TK>
__global__ __launch_bounds__(WG_THREADS, 4) void wgrad_kernel(const WgradParams p) {
  // LDS ring depth: the loop is bounded by bytes in flight per CU (L2/HBM latency x fill rate), so use all the
  // LDS two workgroups can share: 6 x 12 KiB (TK=64) or 4 x 16 KiB (TK=128) per workgroup.
  constexpr int NS = TK == 64 ? 6 : 4;
  constexpr int KT = TK / 64;               // 16-col k tiles per wave (wave tile = 64 n x TK/4 k)
  constexpr int YB = TN * 2, XB = TK * 2;   // row bytes
  constexpr int Y_STAGE = MC * YB;          // 8 KiB
  constexpr int X_STAGE = MC * XB;          // 8 | 4 KiB
  constexpr int STAGE = Y_STAGE + X_STAGE;
  constexpr int X_WAVES = X_STAGE / 1024;   // waves that own an X piece (8 | 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const int ntile = p.tiles_n * p.tiles_k;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);   // one split's tiles run on one XCD: slab rows re-read from its L2
  const int split = lid / ntile, tile = lid % ntile;
  const int n0 = (tile / p.tiles_k) * TN, k0 = (tile % p.tiles_k) * TK;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const bool do_bias = p.bslab != nullptr && (tile % p.tiles_k) == 0 && wk == 0;
  const bool x_owner = wave < X_WAVES;

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

  // this wave's DMA pieces: Y rows [4w, 4w+4) (1 KiB), X rows [4w, 4w+4) (TK=128) or [8w, 8w+8) (TK=64, waves 0-3)
  const int yrow = wave * 4 + (lane >> 4), yc16 = lane & 15;
  const int ycol = (((yc16 >> 1) ^ swz32<YB>(yrow)) << 4) + ((yc16 & 1) << 3);
  const int ygn = min(n0 + ycol, p.N - 8);
  constexpr int XL = XB / 16;                 // 16-byte chunks per X row (16 | 8)
  const int xrow = wave * (1024 / XB) + lane / XL, xc16 = lane % XL;
  const int xcol = (((xc16 >> 1) ^ swz32<XB>(xrow)) << 4) + ((xc16 & 1) << 3);
  const int xgk = min(k0 + xcol, p.K - 8);
  auto issue = [&](int s) {
    unsigned char* st = smem + (s % NS) * STAGE;
    const int mb = mbeg + s * MC;
    const long gy = min(mb + yrow, p.M - 1);
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p.Y + gy * p.ldy + ygn), (lds_void_t*)(st + wave * 1024), 16, 0, 0);
    if (x_owner) {
      const long gx = min(mb + xrow, p.M - 1);
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p.X + gx * p.ldx + xgk), (lds_void_t*)(st + Y_STAGE + wave * 1024), 16, 0, 0);
    }
  };

  const int ns = (mend - mbeg + MC - 1) / MC;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < ns) issue(s);
  for (int s = 0; s < ns; ++s) {
    const int younger = min(NS - 2, ns - 1 - s);      // stages issued after stage s and still allowed in flight
    const int outstanding = younger * (x_owner ? 2 : 1);
    switch (outstanding) {                            // s_waitcnt takes an immediate
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();        // stage s landed everywhere; everyone finished reading stage s-1
    asm volatile("" ::: "memory");
    if (s + NS - 1 < ns) issue(s + NS - 1);
    unsigned char* Ys = smem + (s % NS) * STAGE;
    unsigned char* Xs = Ys + Y_STAGE;
    const int valid = mend - (mbeg + s * MC);          // rows of this stage inside [mbeg, mend)
    if (valid < MC) {                                  // only the last stage of the last split: zero the clamped rows
      for (int i = tid; i < (MC - valid) * (YB / 16); i += WG_THREADS)
        *reinterpret_cast<f32x4*>(Ys + valid * YB + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int i = tid; i < (MC - valid) * (XB / 16); i += WG_THREADS)
        *reinterpret_cast<f32x4*>(Xs + valid * XB + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      __syncthreads();
    }
    bf16x8 af[4], bfr[KT];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tr_frag<YB>(Ys, 0, wn * 64 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < KT; ++j) bfr[j] = tr_frag<XB>(Xs, 0, wk * (TK / 4) + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < KT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
    }
  }

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

// out[i] (+)= sum_s slab[s][i].  Block = 64 float4 columns x 4 split slices (coalesced 1 KiB rows, 4x the
// loads in flight of a one-thread-per-column loop), slices combined through LDS in fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, long n, int splits,
                                                           float* __restrict__ out, int accumulate) {
  __shared__ f32x4 part[4][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + col) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i + 4 <= n) {
#pragma unroll 4
    for (int sp = sl; sp < splits; sp += 4) s += *reinterpret_cast<const f32x4*>(slab + (long)sp * n + i);
  } else if (i < n) {
    for (int sp = sl; sp < splits; sp += 4)
      for (int e = 0; e < 4 && i + e < n; ++e) s[e] += slab[(long)sp * n + i + e];
  }
  part[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    f32x4 t = part[0][col] + part[1][col] + part[2][col] + part[3][col];
    if (i + 4 <= n) {
      f32x4* o = reinterpret_cast<f32x4*>(out + i);
      *o = accumulate ? *o + t : t;
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = accumulate ? out[i + e] + t[e] : t[e];
    }
  }
}

struct WgradPlan { int tk, tiles_n, tiles_k, splits, rows_per_split; };

inline WgradPlan wgrad_plan(int M, int N, int K) {
  WgradPlan w;
  w.tk = (K % 128 == 0 || K > 512) ? 128 : 64;
  w.tiles_n = (N + TN - 1) / TN;
  w.tiles_k = (K + w.tk - 1) / w.tk;
  const int tiles = w.tiles_n * w.tiles_k;
  // Two 8-wave workgroups fit a CU: fill the 512 slots ONCE.  One workgroup more than 512 costs a whole
  // extra round (522 WGs ran 1.5x slower than 504 on the N=768,K=192 shape).
  int splits = 512 / tiles;
  const int max_splits = (M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (M + splits - 1) / splits;
  rps = ((rps + 63) / 64) * 64;
  w.rows_per_split = rps;
  w.splits = (M + rps - 1) / rps;
  return w;
}

}  // namespace

extern "C" size_t iq_wgrad_ws_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  WgradPlan w = wgrad_plan(M, N, K);
  size_t nk = ((size_t)N * K + 3) / 4 * 4;
  return ((size_t)w.splits * nk + (size_t)w.splits * ((N + 3) / 4 * 4)) * sizeof(float);
}

extern "C" int iq_gemm_bf16_wgrad(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dbias, int M,
                                  int N, int K, float* ws, size_t ws_bytes, int accumulate, iq_stream_t stream) {
  if (N <= 0 || K <= 0) return IQ_OK;
  if (!dY || !X || !dW || !ws || M <= 0) return IQ_ERR_ARG;
  if ((N % 8) || (K % 8) || (ldy % 8) || (ldx % 8)) return IQ_ERR_UNSUPPORTED;
  if (ws_bytes < iq_wgrad_ws_bytes(M, N, K)) return IQ_ERR_ARG;
  if (((uintptr_t)dW & 15) != 0) return IQ_ERR_ARG;
  WgradPlan w = wgrad_plan(M, N, K);
  WgradParams p;
  p.Y = (const bf16*)dY; p.X = (const bf16*)X; p.ldy = ldy; p.ldx = ldx; p.M = M; p.N = N; p.K = K;
  const size_t nk = ((size_t)N * K + 3) / 4 * 4;
  p.slab = ws;
  p.bslab = dbias ? ws + (size_t)w.splits * nk : nullptr;
  p.tiles_n = w.tiles_n; p.tiles_k = w.tiles_k; p.splits = w.splits; p.rows_per_split = w.rows_per_split;
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_WGRAD, st);
  const int grid = w.tiles_n * w.tiles_k * w.splits;
  // slab stride must equal N*K for the reduce kernel; nk padding only affects the bias slab offset
  WgradParams q = p;
  if (w.tk == 128) {
    const size_t lds = (size_t)4 * MC * (TN + 128) * 2;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)wgrad_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    wgrad_kernel<128><<<grid, WG_THREADS, lds, st>>>(q);
  } else {
    const size_t lds = (size_t)6 * MC * (TN + 64) * 2;
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    wgrad_kernel<64><<<grid, WG_THREADS, lds, st>>>(q);
  }
  const long n = (long)N * K;
  wgrad_reduce_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(p.slab, n, w.splits, dW, accumulate);
  if (dbias) wgrad_reduce_kernel<<<(N + 255) / 256, 256, 0, st>>>(p.bslab, N, w.splits, dbias, accumulate);
  return iq_launch_status();
}
