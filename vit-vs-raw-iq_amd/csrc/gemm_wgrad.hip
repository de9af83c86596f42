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
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int WG_THREADS = 512;   // 8 waves: 2 (n) x 4 (k); two workgroups per CU hide the HBM latency of the single-stage loop
constexpr int TN = 128;     // output rows (n) per tile
constexpr int MC = 64;      // contraction rows per LDS stage
constexpr int YLD = TN + 16;  // padded LDS row (elements): 288 B rows -> conflict-free tr reads

struct WgradParams {
  const bf16* Y; const bf16* X;
  int ldy, ldx, M, N, K;
  float* slab;       // [splits][N*K]
  float* bslab;      // [splits][N] or null
  int tiles_n, tiles_k, splits, rows_per_split;
};

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

template <int TK>
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

  if (mbeg < mend) {
    gload(mbeg);
    lstore();
  }
  __syncthreads();
  for (int mb = mbeg; mb < mend; mb += MC) {
    const bool more = mb + MC < mend;
    if (more) gload(mb + MC);
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
    __syncthreads();
    if (more) {
      lstore();
      __syncthreads();
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
                                                           float* __restrict__ out, const float* __restrict__ bslab,
                                                           long nb, float* __restrict__ bout, int accumulate) {
  // blocks [0, nblk_w) reduce the weight slabs, the remaining ones the bias slabs (one launch for both)
  __shared__ f32x4 part[4][64];
  const long nblk_w = (n + 255) / 256;
  if ((long)blockIdx.x >= nblk_w) { slab = bslab; n = nb; out = bout; }
  const long blk = (long)blockIdx.x >= nblk_w ? (long)blockIdx.x - nblk_w : (long)blockIdx.x;
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = (blk * 64 + col) * 4;
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
    const size_t lds = (size_t)MC * (YLD + 128 + 16) * 2;
    wgrad_kernel<128><<<grid, WG_THREADS, lds, st>>>(q);
  } else {
    const size_t lds = (size_t)MC * (YLD + 64 + 16) * 2;
    wgrad_kernel<64><<<grid, WG_THREADS, lds, st>>>(q);
  }
  const long n = (long)N * K;
  const int nblk_w = (int)((n + 255) / 256), nblk_b = dbias ? (N + 255) / 256 : 0;
  wgrad_reduce_kernel<<<nblk_w + nblk_b, 256, 0, st>>>(p.slab, n, w.splits, dW, p.bslab, (long)N, dbias, accumulate);
  return iq_launch_status();
}
