import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, D, K, pdrop):
    g = torch.Generator(device="cuda").manual_seed(M + D + K)
    bf = lambda t: t.to(torch.bfloat16)
    A = bf(torch.randn(M, K, device=d, generator=g)); W = bf(torch.randn(D, K, device=d, generator=g) / math.sqrt(K))
    R = bf(torch.randn(M, D, device=d, generator=g)); z = bf(torch.randn(M, D, device=d, generator=g) * 1.5 + 0.3)
    gamma = torch.rand(D, device=d, generator=g) + 0.5
    zf = z.float(); mean = zf.mean(-1).contiguous(); rstd = (1.0 / torch.sqrt(zf.var(-1, unbiased=False) + 1e-12)).contiguous()
    dr = N.Dropout(); dr.seed, dr.step, dr.site, dr.p = 4321, 5, 7, pdrop
    e = N.Epilogue(); e.residual = R.data_ptr(); e.ldr = D
    dx = torch.empty(M, D, dtype=torch.bfloat16, device=d)
    L.iq_gemm_bf16_nt(A.data_ptr(), K, W.data_ptr(), K, dx.data_ptr(), D, M, D, K, C.byref(e), st())
    dz0 = torch.empty_like(z); dy0 = torch.zeros_like(z); dg0 = torch.empty(D, device=d); db0 = torch.empty(D, device=d)
    ws0 = torch.empty(L.iq_ln_bwd_ws_bytes(D), dtype=torch.uint8, device=d)
    L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz0.data_ptr(), dy0.data_ptr(),
                C.byref(dr) if pdrop > 0 else None, dg0.data_ptr(), db0.data_ptr(), ws0.data_ptr(), 0, M, D, st())
    dz1 = torch.full_like(z, float("nan")); dy1 = torch.zeros_like(z)
    rows = L.iq_gemm_lnbwd_partial_rows(M); part = torch.zeros(rows, 2 * D, device=d)
    rc = L.iq_gemm_bf16_lnbwd(A.data_ptr(), K, W.data_ptr(), K, R.data_ptr(), D, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                              gamma.data_ptr(), C.byref(dr) if pdrop > 0 else None, dz1.data_ptr(), dy1.data_ptr(), part.data_ptr(), M, D, K, st())
    torch.cuda.synchronize()
    ref = (A.double() @ W.double().t() + R.double())
    bad = (dz0.view(torch.int16) != dz1.view(torch.int16))
    rows_bad = bad.any(1).nonzero().flatten()
    print(f"M={M} D={D} K={K} p={pdrop}: rc={rc} differing elements {int(bad.sum())} in {rows_bad.numel()} rows; first rows {rows_bad[:8].tolist()}; "
          f"max |dz0-dz1| {(dz0.float()-dz1.float()).abs().max().item():.4g}; dx(2-launch) vs fp64 {(dx.double()-ref).abs().max().item():.4g}")
    if rows_bad.numel():
        r = int(rows_bad[0]); cols = bad[r].nonzero().flatten()[:6].tolist()
        print("  row", r, "cols", cols, "dz0", dz0[r, cols].tolist(), "dz1", dz1[r, cols].tolist())
for M, D, K, p in ((50432, 128, 64, 0.1), (50432, 128, 64, 0.0), (50432, 128, 384, 0.1), (4096, 128, 64, 0.0), (50432, 192, 64, 0.0)):
    run(M, D, K, p)
