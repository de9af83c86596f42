"""GPU (MI355X): every C-ABI op of include/iqvit.h against a plain PyTorch fp32/fp64 reference of the
same op, called through ctypes exactly as the host layer calls it.  Inputs are rounded to bf16 first so
the comparison isolates the kernel (fp32 accumulate, bf16 store) from input quantisation.

Tolerances (stated, floating point): bf16 outputs  |err| <= 2^-7 * |ref| + small abs (one bf16 ulp is
2^-8 relative);  fp32 outputs (statistics, weight gradients, losses) rtol 2e-3 of the tensor scale,
which is the bf16-product / fp32-accumulate error at these contraction lengths.
"""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import vit_vs_raw_iq_amd._native as N
    return N.lib()


def _N():
    import vit_vs_raw_iq_amd._native as N
    return N


def dev():
    return torch.device("cuda:0")


def bf(t):
    return t.to(torch.bfloat16)


def close_bf16(got, ref, what="", rel=2 ** -7, abs_=None):
    got = got.float()
    ref = ref.float()
    scale = ref.abs().max().item() + 1e-12
    abs_ = abs_ if abs_ is not None else 4e-3 * scale
    err = (got - ref).abs()
    bad = err > (rel * ref.abs() + abs_)
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, max err {err.max().item():.4g} (scale {scale:.4g})"


def close_f32(got, ref, what="", rtol=2e-3):
    scale = ref.abs().max().item() + 1e-12
    err = (got.float() - ref.float()).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


def stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [64, 128, 192, 256, 768, 32, 96, 136, 176, 1048])
@pytest.mark.parametrize("M", [1, 37, 4100])
def test_ln_fwd_bwd(L, D, M):
    N = _N()
    assert L.iq_ln_supported(D) == 1
    g = torch.Generator(device="cuda").manual_seed(D * 1000 + M)
    z = bf(torch.randn(M, D, device=dev(), generator=g) * 2 + 0.3)
    gamma = torch.randn(D, device=dev(), generator=g)
    beta = torch.randn(D, device=dev(), generator=g)
    x = torch.empty_like(z)
    mean = torch.empty(M, device=dev())
    rstd = torch.empty(M, device=dev())
    N.check(L.iq_ln_fwd(z.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.data_ptr(), mean.data_ptr(),
                        rstd.data_ptr(), M, D, 1e-12, stream()), "ln_fwd")
    zf = z.double()
    mu = zf.mean(-1, keepdim=True)
    var = zf.var(-1, unbiased=False, keepdim=True)
    ref = gamma.double() * ((zf - mu) / torch.sqrt(var + 1e-12)) + beta.double()
    close_bf16(x, ref, "ln out")
    close_f32(mean, mu.squeeze(1), "mean", 1e-5)
    close_f32(rstd, (1 / torch.sqrt(var + 1e-12)).squeeze(1), "rstd", 1e-4)

    dx = bf(torch.randn(M, D, device=dev(), generator=g))
    dz = torch.empty_like(z)
    dgamma = torch.full((D,), 7.0, device=dev())
    dbeta = torch.full((D,), -3.0, device=dev())
    ws = torch.empty(L.iq_ln_bwd_ws_bytes(D), dtype=torch.uint8, device=dev())
    N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                        dz.data_ptr(), None, None, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), 0, M, D,
                        stream()), "ln_bwd")
    zr = z.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    mu = zr.mean(-1, keepdim=True)
    var = zr.var(-1, unbiased=False, keepdim=True)
    out = gr * ((zr - mu) / torch.sqrt(var + 1e-12)) + br
    out.backward(dx.double())
    close_bf16(dz, zr.grad, "ln dz")
    close_f32(dgamma, gr.grad, "dgamma", 1e-3)
    close_f32(dbeta, br.grad, "dbeta", 1e-3)
    # accumulate=1 adds
    N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                        dz.data_ptr(), None, None, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), 1, M, D,
                        stream()), "ln_bwd acc")
    close_f32(dgamma, 2 * gr.grad, "dgamma acc", 1e-3)


def test_ln_unsupported_width_is_refused(L):
    assert L.iq_ln_supported(20) == 0       # rows must be whole 16-byte vectors
    assert L.iq_ln_supported(136) == 1      # 17 vectors per row: masked one-wave-per-row fallback
    assert L.iq_ln_supported(4096) == 0


# ------------------------------------------------------------------------------------------------
# GEMM NT + epilogues
# ------------------------------------------------------------------------------------------------
def run_gemm(L, A, B, M, N_, K, **kw):
    N = _N()
    Cout = torch.zeros(kw.pop("rows_out", M), N_, dtype=torch.bfloat16, device=dev())
    e = N.Epilogue()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            setattr(e, k, v.data_ptr())
        elif k == "drop":
            e.drop = v
        else:
            setattr(e, k, v)
    N.check(L.iq_gemm_bf16_nt(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), Cout.data_ptr(), N_, M, N_, K,
                              C.byref(e), stream()), "gemm_nt")
    torch.cuda.synchronize()
    return Cout


@pytest.mark.parametrize("M,N_,K", [(1, 64, 32), (130, 192, 192), (1000, 576, 192), (777, 768, 192), (300, 192, 768),
                                    (513, 128, 1024), (257, 40, 72), (129, 384, 128), (2000, 1024, 256)])
def test_gemm_plain_and_bias_relu(L, M, N_, K):
    g = torch.Generator(device="cuda").manual_seed(M + N_ + K)
    A = bf(torch.randn(M, K, device=dev(), generator=g))
    B = bf(torch.randn(N_, K, device=dev(), generator=g) / math.sqrt(K))
    bias = torch.randn(N_, device=dev(), generator=g)
    ref = A.double() @ B.double().t()
    close_bf16(run_gemm(L, A, B, M, N_, K), ref, "plain")
    close_bf16(run_gemm(L, A, B, M, N_, K, bias=bias), ref + bias.double(), "bias")
    close_bf16(run_gemm(L, A, B, M, N_, K, bias=bias, relu=1), torch.relu(ref + bias.double()), "bias+relu")


@pytest.mark.parametrize("M,N_,K", [(16384, 576, 192), (16384 + 77, 768, 192), (20000, 512, 128), (50432, 768, 192),
                                    (50432, 576, 192), (50001, 768, 64), (100003, 512, 128), (41472 + 5, 320, 192),
                                    (57344, 256, 160), (49664, 1024, 128), (49663, 1024, 128)])
def test_gemm_wide_n_many_rows(L, M, N_, K):
    """Wide N, K <= 192, many rows -- the cfg B FFN1 / QKV / FFN2-dgrad shapes and their neighbours (half column tile at
    N = 576 / 320, K = 64 / 128 / 160, 100 k rows): ragged M tail, strided A, bias+ReLU,
    gate, residual, and a dropout mask that depends only on (seed, step, site, element index), not on M."""
    g = torch.Generator(device="cuda").manual_seed(M + N_)
    Abig = bf(torch.randn(M, K + 8, device=dev(), generator=g))
    A = Abig[:, :K]
    B = bf(torch.randn(N_, K, device=dev(), generator=g) / math.sqrt(K))
    bias = torch.randn(N_, device=dev(), generator=g)
    ref = A.double() @ B.double().t()
    close_bf16(run_gemm(L, A, B, M, N_, K), ref, "plain")
    close_bf16(run_gemm(L, A, B, M, N_, K, bias=bias, relu=1), torch.relu(ref + bias.double()), "bias+relu")
    G = bf(torch.randn(M, N_, device=dev(), generator=g)); R = bf(torch.randn(M, N_, device=dev(), generator=g))
    out = run_gemm(L, A, B, M, N_, K, gate=G, ldg=N_, gate_scale=1.25)
    close_bf16(out, torch.where(G.double() > 0, ref * 1.25, torch.zeros_like(ref)), "gate")
    close_bf16(run_gemm(L, A, B, M, N_, K, residual=R, ldr=N_), ref + R.double(), "residual")
    dr = _drop(99, 3, 5, 0.25)
    out = run_gemm(L, A, B, M, N_, K, bias=bias, relu=1, drop=dr)
    keep = out.float() != 0
    full = torch.relu(ref + bias.double())
    frac = keep.float().mean().item() / (full != 0).float().mean().item()
    assert abs(frac - 0.75) < 0.01, frac
    close_bf16(out.float()[keep], (full / 0.75).float()[keep], "dropout scale")
    # the mask depends only on (seed, step, site, element index): a small-M call (tiled kernel) must reproduce its rows
    Ms = 1000
    out_s = run_gemm(L, A[:Ms], B, Ms, N_, K, bias=bias, relu=1, drop=dr)
    sel = full[:Ms].abs() > 0.05                       # away from the ReLU edge, where summation order could flip a sign
    assert torch.equal((out_s.float() != 0)[sel], keep[:Ms][sel])


@pytest.mark.parametrize("M,N_,K", [(50432, 768, 768), (40192, 1024, 256), (40000 + 77, 1024, 256), (100864, 768, 3072), (35072, 2304, 768),
                                    (65536, 512, 320), (50000, 768, 768)])
def test_gemm_big_tiles(L, M, N_, K):
    """The MFMA-bound shapes (ViT-Base: D 768, F 3072, 512 frames x 197 tokens) run the persistent 256 x 256-tile kernel
    (gemm_big.hip): every epilogue it serves, 4 to 48 K-tiles, an odd K-tile count (K = 320), ragged last row blocks
    (40077 and 50000 rows), and the same dropout mask as the tiled kernel."""
    g = torch.Generator(device="cuda").manual_seed(M + N_ + K)
    A = bf(torch.randn(M, K, device=dev(), generator=g))
    B = bf(torch.randn(N_, K, device=dev(), generator=g) / math.sqrt(K))
    bias = torch.randn(N_, device=dev(), generator=g)
    ref = (A.float() @ B.float().t()).double() if K > 1024 else A.double() @ B.double().t()
    close_bf16(run_gemm(L, A, B, M, N_, K), ref, "plain")
    close_bf16(run_gemm(L, A, B, M, N_, K, bias=bias, relu=1), torch.relu(ref + bias.double()), "bias+relu")
    G = bf(torch.randn(M, N_, device=dev(), generator=g)); R = bf(torch.randn(M, N_, device=dev(), generator=g))
    out = run_gemm(L, A, B, M, N_, K, gate=G, ldg=N_, gate_scale=1.25)
    close_bf16(out, torch.where(G.double() > 0, ref * 1.25, torch.zeros_like(ref)), "gate")
    del out
    close_bf16(run_gemm(L, A, B, M, N_, K, bias=bias, residual=R, ldr=N_), ref + bias.double() + R.double(), "bias+residual")
    dr = _drop(7, 11, 3, 0.1)
    out = run_gemm(L, A, B, M, N_, K, bias=bias, drop=dr, residual=R, ldr=N_)
    pre = ref + bias.double()
    kept = out != R                                     # a dropped element is the residual, exactly
    frac = kept.float().mean().item()
    assert abs(frac - 0.9) < 0.01, frac
    close_bf16(out.float()[kept], (pre / 0.9 + R.double()).float()[kept], "dropout+residual")
    # same (seed, step, site, element) mask as the tiled kernel on a row range small enough to stay on it
    Ms = 1500
    out_s = run_gemm(L, A[:Ms], B, Ms, N_, K, bias=bias, drop=dr, residual=R[:Ms], ldr=N_)
    sel = pre[:Ms].abs() > 0.05
    assert torch.equal((out_s != R[:Ms])[sel], kept[:Ms][sel])


@pytest.mark.parametrize("D,K", [(192, 192), (192, 768), (128, 128), (128, 1024), (256, 256), (256, 1024), (128, 64)])
@pytest.mark.parametrize("M,pdrop", [(1000, 0.0), (50432, 0.1), (130, 0.25), (77, 0.0), (16640, 0.2)])   # 16,640 = cfg C at 256 frames: 64-row blocks
def test_gemm_with_fused_layernorm_tail_equals_gemm_then_layernorm(L, D, K, M, pdrop):
    """iq_gemm_bf16_ln (out-projection / FFN2 + dropout + residual + LayerNorm in one launch, encoder_layer.py:24-25,32-33)
    against the two-kernel path it replaces (iq_gemm_bf16_nt with the same epilogue, then iq_ln_fwd): Z bit for bit
    (same MFMA order, same Philox mask), statistics to fp32 summation order, X within one bf16 ulp; and both against fp64."""
    N = _N()
    assert L.iq_gemm_ln_supported(D, K) == 1 and L.iq_gemm_ln_supported(768, 768) == 0 and L.iq_gemm_ln_supported(192, 40) == 0
    g = torch.Generator(device="cuda").manual_seed(M + D + K)
    A = bf(torch.randn(M, K, device=dev(), generator=g))
    W = bf(torch.randn(D, K, device=dev(), generator=g) / math.sqrt(K))
    R = bf(torch.randn(M, D, device=dev(), generator=g) + 0.2)
    bias = torch.randn(D, device=dev(), generator=g)
    gamma = torch.rand(D, device=dev(), generator=g) + 0.5
    beta = torch.randn(D, device=dev(), generator=g)
    dr = _drop(1234, 7, 4, pdrop)
    # two-kernel path
    Z0 = run_gemm(L, A, W, M, D, K, bias=bias, drop=dr, residual=R, ldr=D) if pdrop > 0 else run_gemm(L, A, W, M, D, K, bias=bias, residual=R, ldr=D)
    X0 = torch.empty_like(Z0)
    mean0 = torch.empty(M, device=dev()); rstd0 = torch.empty(M, device=dev())
    N.check(L.iq_ln_fwd(Z0.data_ptr(), gamma.data_ptr(), beta.data_ptr(), X0.data_ptr(), mean0.data_ptr(), rstd0.data_ptr(),
                        M, D, 1e-12, stream()), "ln_fwd")
    # fused
    Z1 = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=dev())
    X1 = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=dev())
    mean1 = torch.full((M,), float("nan"), device=dev()); rstd1 = torch.full((M,), float("nan"), device=dev())
    N.check(L.iq_gemm_bf16_ln(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), R.data_ptr(), D, C.byref(dr) if pdrop > 0 else None,
                              gamma.data_ptr(), beta.data_ptr(), 1e-12, Z1.data_ptr(), X1.data_ptr(), mean1.data_ptr(),
                              rstd1.data_ptr(), M, D, K, stream()), "gemm_ln")
    assert torch.equal(Z0.view(torch.int16), Z1.view(torch.int16)), "Z differs from the two-kernel path"
    assert torch.allclose(mean0, mean1, rtol=0, atol=2e-6 * (mean0.abs().max().item() + 1))
    assert torch.allclose(rstd0, rstd1, rtol=2e-6, atol=0)
    ulp = (X0.float() - X1.float()).abs() / (X0.float().abs() * 2 ** -7 + 1e-6)
    assert ulp.max().item() <= 1.0 + 1e-3, ulp.max().item()
    assert (X0.view(torch.int16) != X1.view(torch.int16)).float().mean().item() < 1e-3      # and almost always identical
    if pdrop == 0.0:
        zf = (A.double() @ W.double().t() + bias.double()) + R.double()
        close_bf16(Z1, zf, "Z vs fp64")
        zb = Z1.double()
        mu = zb.mean(-1, keepdim=True); var = zb.var(-1, unbiased=False, keepdim=True)
        close_bf16(X1, gamma.double() * ((zb - mu) / torch.sqrt(var + 1e-12)) + beta.double(), "X vs fp64")
        close_f32(mean1, mu.squeeze(-1), "mean", 1e-5)
        close_f32(rstd1, (1 / torch.sqrt(var + 1e-12)).squeeze(-1), "rstd", 1e-5)


@pytest.mark.parametrize("D,K", [(192, 768), (192, 576), (128, 1024), (128, 384), (128, 64)])
@pytest.mark.parametrize("M,pdrop", [(1000, 0.0), (50432, 0.1), (130, 0.25), (77, 0.0), (16640, 0.2)])
def test_dgrad_with_fused_layernorm_backward_equals_gemm_then_ln_bwd(L, D, K, M, pdrop):
    """iq_gemm_bf16_lnbwd (FFN1 / QKV data gradient + residual + the LayerNorm backward that consumes it, one launch)
    against the two launches it replaces (iq_gemm_bf16_nt with the residual, then iq_ln_bwd): dZ and dY bit for bit
    (the rounded dX tile crosses LDS instead of HBM, same arithmetic order), gamma / beta gradients after their
    fixed-order reduction to fp32 summation order; and against fp64 torch autograd of LayerNorm."""
    N = _N()
    assert L.iq_gemm_lnbwd_supported(D, K) == 1 and L.iq_gemm_lnbwd_supported(256, 256) == 0
    g = torch.Generator(device="cuda").manual_seed(M + D + K)
    A = bf(torch.randn(M, K, device=dev(), generator=g))
    W = bf(torch.randn(D, K, device=dev(), generator=g) / math.sqrt(K))
    R = bf(torch.randn(M, D, device=dev(), generator=g))
    z = bf(torch.randn(M, D, device=dev(), generator=g) * 1.5 + 0.3)
    gamma = torch.rand(D, device=dev(), generator=g) + 0.5
    zf = z.float()
    mean = zf.mean(-1).contiguous()
    rstd = (1.0 / torch.sqrt(zf.var(-1, unbiased=False) + 1e-12)).contiguous()
    dr = _drop(4321, 5, 7, pdrop)
    # two launches
    dx = run_gemm(L, A, W, M, D, K, residual=R, ldr=D)
    dz0 = torch.empty_like(z); dy0 = torch.zeros_like(z)
    dg0 = torch.empty(D, device=dev()); db0 = torch.empty(D, device=dev())
    ws0 = torch.empty(L.iq_ln_bwd_ws_bytes(D), dtype=torch.uint8, device=dev())
    N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz0.data_ptr(),
                        dy0.data_ptr(), C.byref(dr) if pdrop > 0 else None, dg0.data_ptr(), db0.data_ptr(), ws0.data_ptr(), 0,
                        M, D, stream()), "ln_bwd")
    # fused
    dz1 = torch.full_like(z, float("nan")); dy1 = torch.zeros_like(z)
    rows = L.iq_gemm_lnbwd_partial_rows(M)
    assert rows == ((M + 127) // 128 if M > 320 * 128 else (M + 63) // 64)      # 64-row blocks while 128-row ones would not fill the chip
    part = torch.full((rows, 2 * D), float("nan"), device=dev())
    N.check(L.iq_gemm_bf16_lnbwd(A.data_ptr(), K, W.data_ptr(), K, R.data_ptr(), D, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 gamma.data_ptr(), C.byref(dr) if pdrop > 0 else None, dz1.data_ptr(), dy1.data_ptr(),
                                 part.data_ptr(), M, D, K, stream()), "gemm_lnbwd")
    def same(a, b, what):
        if K >= 384:        # both paths add the residual inside the MFMA (gemm_nt's RESK tile): bit for bit
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"{what} differs from the two-launch path"
        else:               # the unfused GEMM adds it in the VALU epilogue: a handful of dX elements round the other way at a tie
            rows_off = (a.view(torch.int16) != b.view(torch.int16)).any(1).float().mean().item()
            assert rows_off < 2e-3 and (a.float() - b.float()).abs().max().item() <= 2 ** -6 * a.float().abs().max().item(), (what, rows_off)
    same(dz0, dz1, "dZ")
    if pdrop > 0:
        same(dy0, dy1, "dY")
        kept = (dy1.float() != 0).float().mean().item() / max((dz1.float() != 0).float().mean().item(), 1e-9)
        assert abs(kept - (1 - pdrop)) < 0.02
    dg1, db1 = part[:, :D].sum(0), part[:, D:].sum(0)
    close_f32(dg1, dg0, "dgamma", 1e-4)
    close_f32(db1, db0, "dbeta", 1e-4)
    # against autograd of the LayerNorm itself (fp64) on the same rounded dX
    zz = z.double().requires_grad_(True)
    gg = gamma.double().requires_grad_(True)
    mu = zz.mean(-1, keepdim=True); var = zz.var(-1, unbiased=False, keepdim=True)
    y = gg * ((zz - mu) / torch.sqrt(var + 1e-12))
    y.backward(dx.double())
    close_bf16(dz1, zz.grad, "dZ vs fp64 autograd", abs_=2e-2 * zz.grad.abs().max().item())
    close_f32(dg1, gg.grad, "dgamma vs fp64", 5e-3)


def _epi(N, **kw):
    e = N.Epilogue()
    for k, v in kw.items():
        if k in ("bias", "gate", "residual"):
            setattr(e, k, v.data_ptr())
        elif k == "drop":
            e.drop = v
        else:
            setattr(e, k, v)
    return e


@pytest.mark.parametrize("frames,S,D,F,pdrop", [(256, 197, 192, 768, 0.1), (7, 197, 192, 768, 0.0), (256, 65, 128, 1024, 0.2),
                                                 (5, 5, 128, 512, 0.0), (3, 224, 192, 128, 0.3), (9, 17, 128, 256, 0.1),
                                                 (4, 129, 128, 512, 0.0), (2, 1, 192, 256, 0.0), (100, 197, 192, 64, 0.1),
                                                 (40, 197, 192, 192, 0.1), (520, 65, 128, 1024, 0.2), (400, 65, 128, 512, 0.0)])
def test_ffn_chain_equals_ffn1_then_ffn2_layernorm(L, frames, S, D, F, pdrop):
    """iq_ffn_chain_fwd (the whole feed-forward sub-layer + norm2 of a frame in one workgroup: position_wise_feed_forward.py:12-17,
    encoder_layer.py:30-33) against the two launches it replaces -- iq_gemm_bf16_nt (bias, ReLU, dropout1) then iq_gemm_bf16_ln:
    H and Z bit for bit (same K order, same Philox counters), statistics to fp32 summation order, X within one bf16 ulp; and
    against fp64 without dropout."""
    N = _N()
    assert L.iq_ffn_chain_supported(S, D, F) == 1
    assert L.iq_ffn_chain_supported(S, 256, F) == 0 and L.iq_ffn_chain_supported(S, D, 96) == 0
    M = frames * S
    g = torch.Generator(device="cuda").manual_seed(M + D + F)
    X1 = bf(torch.randn(M, D, device=dev(), generator=g))
    W1 = bf(torch.randn(F, D, device=dev(), generator=g) / math.sqrt(D))
    W2 = bf(torch.randn(D, F, device=dev(), generator=g) / math.sqrt(F))
    b1, b2 = torch.randn(F, device=dev(), generator=g), torch.randn(D, device=dev(), generator=g)
    gamma = torch.rand(D, device=dev(), generator=g) + 0.5
    beta = torch.randn(D, device=dev(), generator=g)
    d1, d2 = _drop(77, 3, 5, pdrop), _drop(77, 3, 6, pdrop)
    # two launches
    H0 = run_gemm(L, X1, W1, M, F, D, bias=b1, relu=1, drop=d1) if pdrop > 0 else run_gemm(L, X1, W1, M, F, D, bias=b1, relu=1)
    Z0 = torch.empty(M, D, dtype=torch.bfloat16, device=dev()); X0 = torch.empty_like(Z0)
    mean0 = torch.empty(M, device=dev()); rstd0 = torch.empty(M, device=dev())
    N.check(L.iq_gemm_bf16_ln(H0.data_ptr(), F, W2.data_ptr(), F, b2.data_ptr(), X1.data_ptr(), D, C.byref(d2) if pdrop > 0 else None,
                              gamma.data_ptr(), beta.data_ptr(), 1e-12, Z0.data_ptr(), X0.data_ptr(), mean0.data_ptr(),
                              rstd0.data_ptr(), M, D, F, stream()), "gemm_ln")
    # one launch
    nan = float("nan")
    H1 = torch.full((M, F), nan, dtype=torch.bfloat16, device=dev())
    Z1 = torch.full((M, D), nan, dtype=torch.bfloat16, device=dev()); X2 = torch.full((M, D), nan, dtype=torch.bfloat16, device=dev())
    mean1 = torch.full((M,), nan, device=dev()); rstd1 = torch.full((M,), nan, device=dev())
    N.check(L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(d1) if pdrop > 0 else None, H1.data_ptr(),
                               W2.data_ptr(), b2.data_ptr(), C.byref(d2) if pdrop > 0 else None, gamma.data_ptr(), beta.data_ptr(),
                               1e-12, Z1.data_ptr(), X2.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(), None, frames, S, D, F, stream()),
            "ffn_chain")
    def ties_only(a, b, what, frac, abs_=1e-5):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert ne.float().mean().item() <= frac, f"{what}: {ne.float().mean().item():.3g} of the elements differ"
        # one bf16 ulp of the larger value, or -- next to a ReLU zero -- an fp32 rounding of the pre-activation
        excess = (a.float() - b.float()).abs() - (torch.maximum(a.float().abs(), b.float().abs()) * 2 ** -7 + abs_)
        assert excess.max().item() <= 0.0, f"{what}: more than one bf16 ulp apart ({excess.max().item():.3g})"
    assert ((H0 == 0) != (H1 == 0)).float().mean().item() <= 1e-4, "ReLU / dropout pattern of H differs from the FFN1 launch"
    ties_only(H0, H1, "H", 2e-4)
    # Z: a one-ulp difference of an H element (|h| ~ 1-4) moves the 192 sums it enters by ~|h| 2^-8 |w2| ~ 1e-3: beside ties,
    # elements with |z| << 1 may land two or three of their own (small) ulps apart
    ties_only(Z0, Z1, "Z", 2e-3, abs_=4e-3)
    same = (Z0.view(torch.int16) == Z1.view(torch.int16)).all(1)            # rows whose Z agrees bit for bit
    assert same.float().mean().item() > 0.7
    assert torch.allclose(mean0[same], mean1[same], rtol=0, atol=2e-6 * (mean0.abs().max().item() + 1))
    assert torch.allclose(rstd0[same], rstd1[same], rtol=2e-6, atol=0)
    ulp = (X0.float() - X2.float()).abs()[same] / (X0.float().abs()[same] * 2 ** -7 + 1e-6)
    assert ulp.max().item() <= 1.0 + 1e-3, ulp.max().item()
    if pdrop > 0:
        kept = (H1.float() != 0).float().mean().item() / max((run_gemm(L, X1, W1, M, F, D, bias=b1, relu=1).float() != 0).float().mean().item(), 1e-9)
        assert abs(kept - (1 - pdrop)) < 0.02, kept
    else:
        h = torch.relu(X1.double() @ W1.double().t() + b1.double()).to(torch.bfloat16).double()
        zf = h @ W2.double().t() + b2.double() + X1.double()
        close_bf16(Z1, zf, "Z vs fp64")
        zb = Z1.double()
        mu = zb.mean(-1, keepdim=True); var = zb.var(-1, unbiased=False, keepdim=True)
        close_bf16(X2, gamma.double() * ((zb - mu) / torch.sqrt(var + 1e-12)) + beta.double(), "X vs fp64")


@pytest.mark.parametrize("with_qkv", [False, True])
@pytest.mark.parametrize("frames,S,D,F,pdrop", [(256, 197, 192, 768, 0.1), (7, 197, 192, 832, 0.0), (256, 65, 128, 1024, 0.2),
                                                 (5, 5, 128, 512, 0.0), (40, 197, 192, 64, 0.1), (2, 1, 192, 256, 0.0),
                                                 (130, 65, 128, 128, 0.1), (520, 65, 128, 1024, 0.2), (400, 65, 128, 512, 0.0)])
def test_attn_out_ffn_chain_equals_gemm_ln_then_ffn_chain(L, frames, S, D, F, pdrop, with_qkv):
    """iq_attn_out_ffn_chain_fwd (encoder_layer.py:24-33 from the attention output on, one launch) against the two launches it
    replaces: its first stage (output projection + dropout + residual + norm1) against iq_gemm_bf16_ln -- equal except at
    rounding ties (k-values enter the MFMAs in another lane-group order), same dropout mask -- its second stage bit for bit
    against iq_ffn_chain_fwd run on the X1 it wrote, and (with_qkv) the next layer's packed q,k,v projection of its output
    (multi_head_attention.py:17-19) against iq_gemm_bf16_nt on the X it wrote."""
    N = _N()
    M = frames * S
    g = torch.Generator(device="cuda").manual_seed(M + D + F + 1)
    A = bf(torch.randn(M, D, device=dev(), generator=g))
    R = bf(torch.randn(M, D, device=dev(), generator=g))
    Wo = bf(torch.randn(D, D, device=dev(), generator=g) / math.sqrt(D))
    W1 = bf(torch.randn(F, D, device=dev(), generator=g) / math.sqrt(D))
    W2 = bf(torch.randn(D, F, device=dev(), generator=g) / math.sqrt(F))
    bo, b1, b2 = (torch.randn(n, device=dev(), generator=g) for n in (D, F, D))
    g1, g2 = (torch.rand(D, device=dev(), generator=g) + 0.5 for _ in range(2))
    be1, be2 = (torch.randn(D, device=dev(), generator=g) for _ in range(2))
    d0, d1, d2 = _drop(78, 2, 4, pdrop), _drop(78, 2, 5, pdrop), _drop(78, 2, 6, pdrop)
    dp = lambda d: C.byref(d) if pdrop > 0 else None
    nan = float("nan")
    new = lambda *shape, dt=torch.bfloat16: torch.full(shape, nan, dtype=dt, device=dev())
    # reference first stage
    Z1r, X1r, m1r, r1r = new(M, D), new(M, D), new(M, dt=torch.float32), new(M, dt=torch.float32)
    N.check(L.iq_gemm_bf16_ln(A.data_ptr(), D, Wo.data_ptr(), D, bo.data_ptr(), R.data_ptr(), D, dp(d0), g1.data_ptr(), be1.data_ptr(),
                              1e-12, Z1r.data_ptr(), X1r.data_ptr(), m1r.data_ptr(), r1r.data_ptr(), M, D, D, stream()), "gemm_ln")
    # one launch
    Z1, X1, m1, r1 = new(M, D), new(M, D), new(M, dt=torch.float32), new(M, dt=torch.float32)
    H, Z2, X, m2, r2 = new(M, F), new(M, D), new(M, D), new(M, dt=torch.float32), new(M, dt=torch.float32)
    gate = torch.zeros(L.iq_ffn_chain_gate_bytes(M, F) // 4, dtype=torch.int32, device=dev())
    Wq = bf(torch.randn(3 * D, D, device=dev(), generator=g) / math.sqrt(D))
    bq = torch.randn(3 * D, device=dev(), generator=g)
    Yq = new(M, 3 * D)
    qkv = (Wq.data_ptr(), bq.data_ptr(), Yq.data_ptr()) if with_qkv else (None, None, None)
    N.check(L.iq_attn_out_ffn_chain_fwd(A.data_ptr(), Wo.data_ptr(), bo.data_ptr(), dp(d0), R.data_ptr(), g1.data_ptr(), be1.data_ptr(),
                                        Z1.data_ptr(), X1.data_ptr(), m1.data_ptr(), r1.data_ptr(), W1.data_ptr(), b1.data_ptr(), dp(d1),
                                        H.data_ptr(), W2.data_ptr(), b2.data_ptr(), dp(d2), g2.data_ptr(), be2.data_ptr(), 1e-12,
                                        Z2.data_ptr(), X.data_ptr(), m2.data_ptr(), r2.data_ptr(), gate.data_ptr(), *qkv, frames, S, D, F,
                                        stream()), "attn_out_ffn_chain")
    for t in (Z1, X1, m1, r1, H, Z2, X, m2, r2) + ((Yq,) if with_qkv else ()):
        assert torch.isfinite(t.float()).all()
    if with_qkv:       # the next layer's q,k,v of the X written: iq_gemm_bf16_nt's result up to rounding ties
        Yr = run_gemm(L, X, Wq, M, 3 * D, D, bias=bq)
        neq = Yq.view(torch.int16) != Yr.view(torch.int16)
        assert neq.float().mean().item() <= 2e-3, neq.float().mean().item()
        exq = (Yq.float() - Yr.float()).abs() - (torch.maximum(Yq.float().abs(), Yr.float().abs()) * 2 ** -7 + 1e-5)
        assert exq.max().item() <= 0.0, exq.max().item()
        close_bf16(Yq, X.double() @ Wq.double().t() + bq.double(), "q,k,v vs fp64")
    ne = Z1.view(torch.int16) != Z1r.view(torch.int16)
    assert ne.float().mean().item() <= 2e-3, ne.float().mean().item()
    excess = (Z1.float() - Z1r.float()).abs() - (torch.maximum(Z1.float().abs(), Z1r.float().abs()) * 2 ** -7 + 1e-5)
    assert excess.max().item() <= 0.0, excess.max().item()
    same = ~ne.any(1)
    assert same.float().mean().item() > 0.7
    assert torch.allclose(m1[same], m1r[same], rtol=0, atol=2e-6 * (m1r.abs().max().item() + 1))
    assert torch.allclose(r1[same], r1r[same], rtol=2e-6, atol=0)
    ulp = (X1.float() - X1r.float()).abs()[same] / (X1r.float().abs()[same] * 2 ** -7 + 1e-6)
    assert ulp.max().item() <= 1.0 + 1e-3, ulp.max().item()
    if pdrop > 0:      # the dropout mask of the projection output: zeros of (Z1 - R) coincide
        assert (((Z1.float() - R.float()) == 0) == ((Z1r.float() - R.float()) == 0)).float().mean().item() > 1 - 1e-3
    # second stage on the X1 the fused launch wrote: identical code path, identical bits
    H2, Z22, X2, m22, r22 = new(M, F), new(M, D), new(M, D), new(M, dt=torch.float32), new(M, dt=torch.float32)
    gate2 = torch.zeros_like(gate)
    N.check(L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), dp(d1), H2.data_ptr(), W2.data_ptr(), b2.data_ptr(), dp(d2),
                               g2.data_ptr(), be2.data_ptr(), 1e-12, Z22.data_ptr(), X2.data_ptr(), m22.data_ptr(), r22.data_ptr(),
                               gate2.data_ptr(), frames, S, D, F, stream()), "ffn_chain")
    assert torch.equal(H.view(torch.int16), H2.view(torch.int16)) and torch.equal(Z2.view(torch.int16), Z22.view(torch.int16))
    assert torch.equal(X.view(torch.int16), X2.view(torch.int16)) and torch.equal(m2, m22) and torch.equal(r2, r22)
    rw = 32 if M > 32768 else 16                       # rows per wave (ffn_chain.hip::chain_shape)
    full = (M // rw) * (F // 64) * 64                  # (bits of rows past M, in the last wave's ragged unit, are never read)
    assert torch.equal(gate[:full], gate2[:full])


@pytest.mark.parametrize("frames,S,D,F,pdrop", [(256, 197, 192, 768, 0.1), (7, 197, 192, 832, 0.0), (256, 65, 128, 1024, 0.2),
                                                 (5, 5, 128, 512, 0.0), (3, 224, 192, 128, 0.3), (9, 17, 128, 896, 0.1),
                                                 (40, 197, 192, 64, 0.1), (2, 1, 192, 256, 0.0), (520, 65, 128, 1024, 0.2),
                                                 (400, 65, 128, 512, 0.0)])
@pytest.mark.parametrize("with_dA", [False, True])
def test_ffn_chain_backward_equals_gate_gemm_then_dgrad_layernorm_backward(L, frames, S, D, F, pdrop, with_dA):
    """iq_ffn_chain_bwd (gate data gradient + FFN1 data gradient + residual + norm1 backward in one launch, gate = the "H > 0"
    bits iq_ffn_chain_fwd leaves) against the two launches it replaces -- iq_gemm_bf16_nt with the gate epilogue on H itself,
    then iq_gemm_bf16_lnbwd: gH / dZ / dY equal except at rounding ties (k-values enter the MFMAs in another lane-group
    order), identical dropout masks, gamma / beta partial sums to fp32 summation order; F / 64 = 12, 13, 14 chunks exercise the
    three phases of the ring in which the tail borrows its scratch.  with_dA: the attention output projection's data gradient
    (autograd of multi_head_attention.py:28) in the same launch against iq_gemm_bf16_nt on the dY written."""
    N = _N()
    M = frames * S
    g = torch.Generator(device="cuda").manual_seed(M + D + F)
    X1 = bf(torch.randn(M, D, device=dev(), generator=g))
    W1 = bf(torch.randn(F, D, device=dev(), generator=g) / math.sqrt(D))
    W2 = bf(torch.randn(D, F, device=dev(), generator=g) / math.sqrt(F))
    b1, b2 = torch.randn(F, device=dev(), generator=g), torch.randn(D, device=dev(), generator=g)
    gamma = torch.rand(D, device=dev(), generator=g) + 0.5
    beta = torch.randn(D, device=dev(), generator=g)
    # forward: H and its gate bits
    H = torch.empty(M, F, dtype=torch.bfloat16, device=dev()); Zf = torch.empty(M, D, dtype=torch.bfloat16, device=dev()); Xf = torch.empty_like(Zf)
    mf = torch.empty(M, device=dev()); rf = torch.empty(M, device=dev())
    gate = torch.zeros(L.iq_ffn_chain_gate_bytes(M, F), dtype=torch.uint8, device=dev())
    d1 = _drop(3, 1, 2, pdrop)
    N.check(L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(d1) if pdrop > 0 else None, H.data_ptr(), W2.data_ptr(),
                               b2.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), 1e-12, Zf.data_ptr(), Xf.data_ptr(), mf.data_ptr(),
                               rf.data_ptr(), gate.data_ptr(), frames, S, D, F, stream()), "ffn_chain_fwd")
    # backward operands
    dO = bf(torch.randn(M, D, device=dev(), generator=g))
    W2t = W2.t().contiguous(); W1t = W1.t().contiguous()                 # [F, D], [D, F]
    R = bf(torch.randn(M, D, device=dev(), generator=g))
    z = bf(torch.randn(M, D, device=dev(), generator=g) * 1.5 + 0.3)
    mean = z.float().mean(-1).contiguous(); rstd = (1.0 / torch.sqrt(z.float().var(-1, unbiased=False) + 1e-12)).contiguous()
    dr = _drop(77, 5, 7, pdrop)
    scale = 65536.0 / (65536.0 - round(pdrop * 65536)) if pdrop > 0 else 1.0
    gH0 = run_gemm(L, dO, W2t, M, F, D, gate=H, ldg=F, gate_scale=scale)
    dz0 = torch.empty_like(z); dy0 = torch.zeros_like(z)
    part0 = torch.empty(L.iq_gemm_lnbwd_partial_rows(M), 2 * D, device=dev())
    N.check(L.iq_gemm_bf16_lnbwd(gH0.data_ptr(), F, W1t.data_ptr(), F, R.data_ptr(), D, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 gamma.data_ptr(), C.byref(dr) if pdrop > 0 else None, dz0.data_ptr(), dy0.data_ptr(), part0.data_ptr(),
                                 M, D, F, stream()), "gemm_lnbwd")
    nan = float("nan")
    gH1 = torch.full((M, F), nan, dtype=torch.bfloat16, device=dev()); dz1 = torch.full_like(z, nan); dy1 = torch.zeros_like(z)
    rows = L.iq_ffn_chain_bwd_partial_rows(M)
    # one row per workgroup: 7 waves of 32 rows above 32,768 rows, 8 | 5 waves of 16 rows below (ffn_chain.hip::chain_shape)
    rw, nw = (32, 7) if M > 32768 else (16, 8) if M > 20480 else (16, 5)
    units = (M + rw - 1) // rw
    assert rows == (units + nw - 1) // nw
    part1 = torch.full((rows, 2 * D), nan, device=dev())
    Wot = bf(torch.randn(D, D, device=dev(), generator=g) / math.sqrt(D))
    dA = torch.full((M, D), nan, dtype=torch.bfloat16, device=dev())
    N.check(L.iq_ffn_chain_bwd(dO.data_ptr(), W2t.data_ptr(), gate.data_ptr(), scale, gH1.data_ptr(), W1t.data_ptr(), R.data_ptr(), z.data_ptr(),
                               mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), C.byref(dr) if pdrop > 0 else None, dz1.data_ptr(),
                               dy1.data_ptr(), part1.data_ptr(), Wot.data_ptr() if with_dA else None, dA.data_ptr() if with_dA else None,
                               frames, S, D, F, stream()), "ffn_chain_bwd")
    if with_dA:        # the output projection's data gradient of the dY (dZ without dropout) this launch wrote
        dAr = run_gemm(L, dy1 if pdrop > 0 else dz1, Wot, M, D, D)
        assert torch.isfinite(dA.float()).all()
        ne = dA.view(torch.int16) != dAr.view(torch.int16)
        assert ne.float().mean().item() <= 3e-3, ne.float().mean().item()
        excess = (dA.float() - dAr.float()).abs() - (torch.maximum(dA.float().abs(), dAr.float().abs()) * 2 ** -7 + 1e-5 * dAr.float().abs().max().item())
        assert excess.max().item() <= 0.0, excess.max().item()
    def ties_only(a, b, what, frac, abs_):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert ne.float().mean().item() <= frac, f"{what}: {ne.float().mean().item():.3g} of the elements differ"
        excess = (a.float() - b.float()).abs() - (torch.maximum(a.float().abs(), b.float().abs()) * 2 ** -7 + abs_)
        assert excess.max().item() <= 0.0, f"{what}: more than one bf16 ulp apart ({excess.max().item():.3g})"
    assert ((gH0 == 0) != (gH1 == 0)).float().mean().item() <= 1e-4, "gate pattern differs"
    ties_only(gH0, gH1, "gH", 2e-4, 1e-5)
    ties_only(dz0, dz1, "dZ", 3e-3, 4e-3 * z.float().abs().max().item())
    if pdrop > 0:
        assert torch.equal(dy0 == 0, dy1 == 0) or ((dy0 == 0) != (dy1 == 0)).float().mean().item() <= 1e-4
        ties_only(dy0, dy1, "dY", 3e-3, 6e-3 * z.float().abs().max().item())
    s0, s1 = part0.sum(0), part1.sum(0)
    assert torch.isfinite(part1).all()
    close_f32(s1, s0, "dgamma | dbeta", 2e-3)


@pytest.mark.parametrize("frames,S,D,F,pdrop", [(256, 197, 192, 768, 0.1), (7, 197, 192, 832, 0.0), (256, 65, 128, 1024, 0.2),
                                                 (5, 5, 128, 512, 0.0), (40, 197, 192, 64, 0.1), (2, 1, 192, 128, 0.0), (9, 17, 128, 896, 0.1),
                                                 (520, 65, 128, 1024, 0.2), (400, 65, 128, 512, 0.0)])
def test_qkv_dgrad_ffn_chain_backward_equals_its_two_launches(L, frames, S, D, F, pdrop):
    """iq_qkv_dgrad_ffn_chain_bwd (q,k,v data gradient of the layer above + norm2 backward + the feed-forward backward + norm1
    backward + output-projection data gradient: encoder_layer.py:24-33 backwards, one launch) against iq_gemm_bf16_lnbwd for
    its first stage (dz2 / dy2 equal except at rounding ties, gamma / beta partial sums to fp32 summation order) and, bit for
    bit, against iq_ffn_chain_bwd run on the dy2 / dz2 it wrote."""
    N = _N()
    M = frames * S
    g = torch.Generator(device="cuda").manual_seed(M + D + F + 2)
    rnd = lambda *shape, s=1.0: bf(torch.randn(*shape, device=dev(), generator=g) * s)
    # forward pass of the feed-forward sub-layer: gate bits
    X1 = rnd(M, D); W1 = rnd(F, D, s=1 / math.sqrt(D)); W2 = rnd(D, F, s=1 / math.sqrt(F))
    b1, b2 = torch.randn(F, device=dev(), generator=g), torch.randn(D, device=dev(), generator=g)
    gam = [torch.rand(D, device=dev(), generator=g) + 0.5 for _ in range(2)]
    beta = torch.randn(D, device=dev(), generator=g)
    nan = float("nan")
    new = lambda *shape, dt=torch.bfloat16: torch.full(shape, nan, dtype=dt, device=dev())
    H, Zf, Xf, mf, rf = new(M, F), new(M, D), new(M, D), new(M, dt=torch.float32), new(M, dt=torch.float32)
    gate = torch.zeros(L.iq_ffn_chain_gate_bytes(M, F), dtype=torch.uint8, device=dev())
    d1 = _drop(3, 1, 2, pdrop)
    dp = lambda d: C.byref(d) if pdrop > 0 else None
    N.check(L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), dp(d1), H.data_ptr(), W2.data_ptr(), b2.data_ptr(), None,
                               gam[0].data_ptr(), beta.data_ptr(), 1e-12, Zf.data_ptr(), Xf.data_ptr(), mf.data_ptr(), rf.data_ptr(),
                               gate.data_ptr(), frames, S, D, F, stream()), "ffn_chain_fwd")
    # backward operands
    gQKV = rnd(M, 3 * D); Wqt = rnd(D, 3 * D, s=1 / math.sqrt(3 * D)); R0 = rnd(M, D)
    W2t = W2.t().contiguous(); W1t = W1.t().contiguous(); Wot = rnd(D, D, s=1 / math.sqrt(D))
    z = [rnd(M, D, s=1.5) + 0.3 for _ in range(2)]                           # z2, z1
    z = [bf(t) for t in z]
    mean = [t.float().mean(-1).contiguous() for t in z]
    rstd = [(1.0 / torch.sqrt(t.float().var(-1, unbiased=False) + 1e-12)).contiguous() for t in z]
    dr2, dr1 = _drop(77, 5, 9, pdrop), _drop(77, 5, 7, pdrop)
    scale = 65536.0 / (65536.0 - round(pdrop * 65536)) if pdrop > 0 else 1.0
    # reference first stage
    dz2r, dy2r = new(M, D), torch.zeros(M, D, dtype=torch.bfloat16, device=dev())
    p2r = torch.empty(L.iq_gemm_lnbwd_partial_rows(M), 2 * D, device=dev())
    N.check(L.iq_gemm_bf16_lnbwd(gQKV.data_ptr(), 3 * D, Wqt.data_ptr(), 3 * D, R0.data_ptr(), D, z[0].data_ptr(), mean[0].data_ptr(),
                                 rstd[0].data_ptr(), gam[0].data_ptr(), dp(dr2), dz2r.data_ptr(), dy2r.data_ptr(), p2r.data_ptr(), M, D,
                                 3 * D, stream()), "gemm_lnbwd")
    # one launch
    rows = L.iq_ffn_chain_bwd_partial_rows(M)
    dz2, dy2, p2 = new(M, D), torch.zeros(M, D, dtype=torch.bfloat16, device=dev()), new(rows, 2 * D, dt=torch.float32)
    gH, dz, dy, dA, p1 = new(M, F), new(M, D), torch.zeros(M, D, dtype=torch.bfloat16, device=dev()), new(M, D), new(rows, 2 * D, dt=torch.float32)
    N.check(L.iq_qkv_dgrad_ffn_chain_bwd(gQKV.data_ptr(), Wqt.data_ptr(), R0.data_ptr(), z[0].data_ptr(), mean[0].data_ptr(), rstd[0].data_ptr(),
                                         gam[0].data_ptr(), dp(dr2), dz2.data_ptr(), dy2.data_ptr(), p2.data_ptr(), W2t.data_ptr(),
                                         gate.data_ptr(), scale, gH.data_ptr(), W1t.data_ptr(), dz2.data_ptr(), z[1].data_ptr(),
                                         mean[1].data_ptr(), rstd[1].data_ptr(), gam[1].data_ptr(), dp(dr1), dz.data_ptr(), dy.data_ptr(),
                                         p1.data_ptr(), Wot.data_ptr(), dA.data_ptr(), frames, S, D, F, stream()), "qkv_dgrad_ffn_chain_bwd")
    for t in (dz2, p2, gH, dz, dA, p1) + ((dy2, dy) if pdrop > 0 else ()):
        assert torch.isfinite(t.float()).all()
    def ties_only(a, b, what, frac, abs_):
        ne = a.view(torch.int16) != b.view(torch.int16)
        assert ne.float().mean().item() <= frac, f"{what}: {ne.float().mean().item():.3g} of the elements differ"
        excess = (a.float() - b.float()).abs() - (torch.maximum(a.float().abs(), b.float().abs()) * 2 ** -7 + abs_)
        assert excess.max().item() <= 0.0, f"{what}: more than one bf16 ulp apart ({excess.max().item():.3g})"
    zmax = z[0].float().abs().max().item()
    ties_only(dz2r, dz2, "dZ2", 3e-3, 4e-3 * zmax)
    if pdrop > 0:
        assert ((dy2r == 0) != (dy2 == 0)).float().mean().item() <= 1e-4
        ties_only(dy2r, dy2, "dY2", 3e-3, 6e-3 * zmax)
    close_f32(p2.sum(0), p2r.sum(0), "dgamma2 | dbeta2", 2e-3)
    # second stage on what the fused launch wrote: identical bits
    gHb, dzb, dyb, dAb, p1b = new(M, F), new(M, D), torch.zeros(M, D, dtype=torch.bfloat16, device=dev()), new(M, D), new(rows, 2 * D, dt=torch.float32)
    dO = dy2 if pdrop > 0 else dz2
    N.check(L.iq_ffn_chain_bwd(dO.data_ptr(), W2t.data_ptr(), gate.data_ptr(), scale, gHb.data_ptr(), W1t.data_ptr(), dz2.data_ptr(), z[1].data_ptr(),
                               mean[1].data_ptr(), rstd[1].data_ptr(), gam[1].data_ptr(), dp(dr1), dzb.data_ptr(), dyb.data_ptr(), p1b.data_ptr(),
                               Wot.data_ptr(), dAb.data_ptr(), frames, S, D, F, stream()), "ffn_chain_bwd")
    for a_, b_, what in ((gH, gHb, "gH"), (dz, dzb, "dZ1"), (dy, dyb, "dY1"), (dA, dAb, "dA")):
        assert torch.equal(a_.view(torch.int16), b_.view(torch.int16)), what
    assert torch.equal(p1, p1b)


@pytest.mark.parametrize("M,K,N_", [(5000, 768, 192), (130, 576, 192), (50432, 768, 192), (999, 384, 192), (4000, 1024, 128),
                                    (700, 384, 128)])
def test_gemm_residual_as_k_stages(L, M, K, N_):
    """C = A W^T + R with N = 192 and nothing else in the tail takes the 192-column tile with the residual streamed as
    extra K stages against identity fragments: must be bit-identical to the ordinary epilogue path (forced here by a
    zero bias, which disables the shortcut) and match fp64."""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = bf(torch.randn(M, K, device=dev(), generator=g))
    B = bf(torch.randn(N_, K, device=dev(), generator=g) / math.sqrt(K))
    Rbig = bf(torch.randn(M, N_ + 64, device=dev(), generator=g))
    R = Rbig[:, :N_]                                                    # ldr > N
    out = run_gemm(L, A, B, M, N_, K, residual=R, ldr=N_ + 64)
    ref = run_gemm(L, A, B, M, N_, K, residual=R, ldr=N_ + 64, bias=torch.zeros(N_, device=dev()))
    # the last fp32 add happens inside the MFMA here and in the VALU there: same value up to the final bf16 rounding,
    # which may break a tie the other way for a handful of elements
    diff = (out.float() - ref.float()).abs()
    assert (diff > 0).float().mean().item() < 1e-4
    assert (diff <= 2.0 ** -7 * ref.float().abs() + 1e-6).all()
    close_bf16(out, A.double() @ B.double().t() + R.double(), "residual-as-K vs fp64")


def test_gemm_asymmetric_layout(L):
    """A = I picks out B^T exactly: catches swapped fragment / C-layout maps (guide 3)."""
    K = 64
    A = bf(torch.eye(K, device=dev()))
    B = bf(torch.arange(128 * K, device=dev()).reshape(128, K).float() % 251 - 125)
    out = run_gemm(L, A, B, K, 128, K)
    assert torch.equal(out.float(), B.float().t())


def test_gemm_residual_gate_strided(L):
    M, N_, K = 515, 192, 576
    g = torch.Generator(device="cuda").manual_seed(5)
    Abig = bf(torch.randn(M, K + 64, device=dev(), generator=g))
    A = Abig[:, :K]                                     # lda > K
    B = bf(torch.randn(N_, K, device=dev(), generator=g) / math.sqrt(K))
    R = bf(torch.randn(M, N_, device=dev(), generator=g))
    G = bf(torch.randn(M, N_, device=dev(), generator=g))
    ref = A.double() @ B.double().t()
    out = run_gemm(L, A, B, M, N_, K, residual=R, ldr=N_)
    close_bf16(out, ref + R.double(), "residual")
    out = run_gemm(L, A, B, M, N_, K, gate=G, ldg=N_, gate_scale=1.25, residual=R, ldr=N_)
    close_bf16(out, torch.where(G.double() > 0, ref * 1.25, torch.zeros_like(ref)) + R.double(), "gate+residual")


def test_gemm_embedding_row_remap_and_pe(L):
    Bf, tok, S, D, K = 7, 9, 10, 64, 32
    g = torch.Generator(device="cuda").manual_seed(9)
    A = bf(torch.randn(Bf * tok, K, device=dev(), generator=g))
    W = bf(torch.randn(D, K, device=dev(), generator=g))
    bias = torch.randn(D, device=dev(), generator=g)
    pe = torch.randn(S, D, device=dev(), generator=g)
    out = run_gemm(L, A, W, Bf * tok, D, K, rows_out=Bf * S, bias=bias, pe=pe, tok=tok, seq=S, cls_off=1)
    ref = (A.double() @ W.double().t() + bias.double()).view(Bf, tok, D) + pe[1:].double()
    out = out.view(Bf, S, D)
    close_bf16(out[:, 1:], ref, "embed rows")
    assert torch.count_nonzero(out[:, 0]) == 0          # cls rows untouched


def _drop(seed, step, site, p):
    N = _N()
    d = N.Dropout()
    d.seed, d.step, d.site, d.p, d.step_dev = seed, step, site, p, None
    return d


def test_dropout_statistics_and_regeneration(L):
    """Philox masks: keep rate, 1/(1-p) scaling, determinism, and the SAME mask regenerated by the LN backward."""
    N = _N()
    M, D = 2048, 192
    A = bf(torch.ones(M, D, device=dev()))
    I = bf(torch.eye(D, device=dev()))
    p = 0.25
    out = run_gemm(L, A, I, M, D, D, drop=_drop(123, 7, 4, p)).float()
    keep = out != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.005
    assert torch.allclose(out[keep], torch.full_like(out[keep], 1 / (1 - p)), rtol=1e-2)
    out2 = run_gemm(L, A, I, M, D, D, drop=_drop(123, 7, 4, p)).float()
    assert torch.equal(out, out2)
    for other in (_drop(124, 7, 4, p), _drop(123, 8, 4, p), _drop(123, 7, 5, p)):
        o = run_gemm(L, A, I, M, D, D, drop=other).float()
        assert (o != 0).ne(keep).float().mean().item() > 0.2
    # device-resident step overrides the host value
    step_dev = torch.tensor([7], dtype=torch.int32, device=dev())
    d = _drop(123, 99, 4, p)
    d.step_dev = step_dev.data_ptr()
    assert torch.equal(run_gemm(L, A, I, M, D, D, drop=d).float(), out)
    # LN backward regenerates the identical mask for dy
    z = bf(torch.randn(M, D, device=dev()))
    gamma = torch.ones(D, device=dev())
    beta = torch.zeros(D, device=dev())
    x = torch.empty_like(z)
    mean = torch.empty(M, device=dev())
    rstd = torch.empty(M, device=dev())
    N.check(L.iq_ln_fwd(z.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                        M, D, 1e-12, stream()), "ln")
    dx = bf(torch.randn(M, D, device=dev()) + 3)
    dz, dy = torch.empty_like(z), torch.empty_like(z)
    dg, db = torch.empty(D, device=dev()), torch.empty(D, device=dev())
    ws = torch.empty(L.iq_ln_bwd_ws_bytes(D), dtype=torch.uint8, device=dev())
    dr = _drop(123, 7, 4, p)
    N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(),
                        dy.data_ptr(), C.byref(dr), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), 0, M, D, stream()), "ln_bwd")
    nz = dz.float() != 0
    assert torch.equal((dy.float() != 0)[nz], keep[nz])
    close_bf16(dy.float()[keep & nz], dz.float()[keep & nz] / (1 - p), "dy scale")


# ------------------------------------------------------------------------------------------------
# weight gradient
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N_,K", [(64, 128, 128), (1000, 192, 192), (5000, 768, 192), (3333, 192, 768),
                                    (777, 576, 192), (130, 40, 72), (4096, 128, 32), (20000, 128, 1024),
                                    (31, 64, 64), (700, 1024, 768),    # N*K > 512 Ki, few rows -> shared-tile kernel
                                    (8192, 768, 768), (16384, 1024, 768), (4096 + 64, 768, 2304), (100864, 768, 768),
                                    (12800, 3072, 768)])               # whole 64-row steps, 256-multiples: gemm_wgrad_big.hip
def test_wgrad(L, M, N_, K):
    N = _N()
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N_ + K)
    dY = bf(torch.randn(M, N_, device=dev(), generator=g))
    X = bf(torch.randn(M, K, device=dev(), generator=g))
    dW = torch.full((N_, K), 5.0, device=dev())
    db = torch.full((N_,), 5.0, device=dev())
    nbytes = L.iq_wgrad_ws_bytes(M, N_, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    N.check(L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N_, K,
                                 ws.data_ptr(), nbytes, 0, stream()), "wgrad")
    ref = dY.double().t() @ X.double()
    close_f32(dW, ref, "dW")
    close_f32(db, dY.double().sum(0), "db")
    N.check(L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N_, K,
                                 ws.data_ptr(), nbytes, 1, stream()), "wgrad acc")
    close_f32(dW, 2 * ref, "dW acc")


@pytest.mark.parametrize("M,shapes,budget", [
    (3940, [(192, 768), (768, 192), (192, 192), (576, 192)], 0),   # one ViT-Tiny encoder layer
    (3940, [(192, 768), (768, 192), (192, 192), (576, 192)], 256), # same with a workgroup budget (overlapped backward)
    (1000, [(128, 256), (40, 72)], 0),                             # partial tiles, two problems
    (515, [(64, 64), (1024, 768), (72, 64)], 0),                   # a large member: falls back to one launch per problem
    (50432, [(192, 768), (768, 192), (192, 192), (576, 192)], 0),  # cfg B's layer at its batch: 256 x 192 LDS-shared tiles (gemm_wgrad_big.hip)
    (16640, [(128, 1024), (1024, 128), (128, 128), (384, 128)], 0),  # cfg C's layer: column tile 128
    (8192, [(768, 3072), (3072, 768), (768, 768), (2304, 768)], 0),  # ViT-Base's layer: column tile 256, one launch for all four
    (4096, [(256, 512), (512, 256)], 0),
])
def test_wgrad_grouped(L, M, shapes, budget):
    """Several weight gradients sharing M in one launch == each one alone (fp64 reference), with and without accumulate."""
    N = _N()
    g = torch.Generator(device="cuda").manual_seed(M)
    probs = (N.WgradProblem * len(shapes))()
    keep, refs = [], []
    pad = 64 if (M % 64 == 0 and M >= 4096) else 8      # (line-aligned rows keep the LDS-shared kernel eligible)
    for i, (n, k) in enumerate(shapes):
        dY = bf(torch.randn(M, n + pad, device=dev(), generator=g))[:, :n]    # ld > N: strided operand
        X = bf(torch.randn(M, k, device=dev(), generator=g))
        dW = torch.full((n, k), 3.0, device=dev())
        db = torch.full((n,), 3.0, device=dev()) if i != 1 else None          # one problem without bias
        keep.append((dY, X, dW, db))
        probs[i].dY = dY.data_ptr(); probs[i].ldy = n + pad; probs[i].X = X.data_ptr(); probs[i].ldx = k
        probs[i].dW = dW.data_ptr(); probs[i].dbias = db.data_ptr() if db is not None else None
        probs[i].N = n; probs[i].K = k
        refs.append((dY.double().t() @ X.double(), dY.double().sum(0)))
    nbytes = L.iq_wgrad_grouped_ws_bytes(probs, len(shapes), M, budget)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    N.check(L.iq_gemm_bf16_wgrad_grouped(probs, len(shapes), M, ws.data_ptr(), nbytes, 0, budget, None, 0, stream()), "grouped")
    for (dY, X, dW, db), (rw, rb) in zip(keep, refs):
        close_f32(dW, rw, "dW")
        if db is not None:
            close_f32(db, rb, "db")
    N.check(L.iq_gemm_bf16_wgrad_grouped(probs, len(shapes), M, ws.data_ptr(), nbytes, 1, budget, None, 0, stream()), "grouped acc")
    for (dY, X, dW, db), (rw, rb) in zip(keep, refs):
        close_f32(dW, 2 * rw, "dW acc")
    assert L.iq_gemm_bf16_wgrad_grouped(probs, len(shapes), M, ws.data_ptr(), nbytes - 1, 0, budget, None, 0, stream()) != 0   # ws too small


@pytest.mark.parametrize("M,D,with_gemm", [(3940, 192, True), (1000, 128, False), (50432, 192, True)])
def test_ln_bwd_partials_reduced_with_the_group(L, M, D, with_gemm):
    """iq_ln_bwd(dgamma=dbeta=NULL) + iq_reduce_seg_t on the grouped launch == iq_ln_bwd reducing on its own (bitwise)."""
    N = _N()
    g = torch.Generator(device="cuda").manual_seed(M + D)
    z = bf(torch.randn(M, D, device=dev(), generator=g)); dx = bf(torch.randn(M, D, device=dev(), generator=g))
    gamma = torch.randn(D, device=dev(), generator=g)
    x = torch.empty_like(z); mean = torch.empty(M, device=dev()); rstd = torch.empty(M, device=dev())
    N.check(L.iq_ln_fwd(z.data_ptr(), gamma.data_ptr(), gamma.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D,
                        1e-12, stream()), "ln_fwd")
    ws = torch.empty(L.iq_ln_bwd_ws_bytes(D), dtype=torch.uint8, device=dev())
    dz = torch.empty_like(z)
    dg_ref, db_ref = torch.empty(D, device=dev()), torch.empty(D, device=dev())
    N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(), None,
                        None, dg_ref.data_ptr(), db_ref.data_ptr(), ws.data_ptr(), 0, M, D, stream()), "ln_bwd")
    dz_ref = dz.clone(); dz.zero_()
    N.check(L.iq_ln_bwd(dx.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(), None,
                        None, None, None, ws.data_ptr(), 0, M, D, stream()), "ln_bwd partial")
    assert torch.equal(dz, dz_ref)
    rows = L.iq_ln_bwd_partial_rows(M, D)
    assert 0 < rows <= 512
    part = ws.view(torch.float32)[: rows * 2 * D].view(rows, 2 * D)
    dg, db = torch.full((D,), 7.0, device=dev()), torch.full((D,), 7.0, device=dev())
    segs = (N.ReduceSeg * 2)()
    for i, (off, out) in enumerate([(0, dg), (D, db)]):
        segs[i].partials = ws.data_ptr() + 4 * off; segs[i].rows = rows; segs[i].row_stride = 2 * D
        segs[i].out = out.data_ptr(); segs[i].n = D
    if with_gemm:
        dY = bf(torch.randn(M, 64, device=dev(), generator=g)); X = bf(torch.randn(M, 64, device=dev(), generator=g))
        dW = torch.empty(64, 64, device=dev())
        pr = (N.WgradProblem * 1)()
        pr[0].dY = dY.data_ptr(); pr[0].ldy = 64; pr[0].X = X.data_ptr(); pr[0].ldx = 64; pr[0].dW = dW.data_ptr()
        pr[0].dbias = None; pr[0].N = 64; pr[0].K = 64
        nb = L.iq_wgrad_grouped_ws_bytes(pr, 1, M, 0); wws = torch.empty(nb, dtype=torch.uint8, device=dev())
        N.check(L.iq_gemm_bf16_wgrad_grouped(pr, 1, M, wws.data_ptr(), nb, 0, 0, segs, 2, stream()), "grouped + extra")
        close_f32(dW, dY.double().t() @ X.double(), "dW")
    else:
        N.check(L.iq_gemm_bf16_wgrad_grouped(None, 0, M, None, 0, 0, 0, segs, 2, stream()), "extra only")
    # same partial rows, fixed-order sums: agree with the stand-alone reduce to rounding (different association)
    close_f32(dg, dg_ref, "dgamma", 1e-5)
    close_f32(db, db_ref, "dbeta", 1e-5)
    close_f32(dg, part[:, :D].double().sum(0), "dgamma vs partial rows", 1e-6)


def test_wgrad_exact_integers(L):
    """Small integers are exact in bf16 and fp32: the transposing LDS reads must reproduce dY^T X bit for bit."""
    N = _N()
    M, N_, K = 200, 128, 64
    dY = bf((torch.arange(M * N_, device=dev()).reshape(M, N_) % 7 - 3).float())
    X = bf((torch.arange(M * K, device=dev()).reshape(M, K) % 5 - 2).float())
    dW = torch.empty(N_, K, device=dev())
    nbytes = L.iq_wgrad_ws_bytes(M, N_, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    N.check(L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), None, M, N_, K, ws.data_ptr(),
                                 nbytes, 0, stream()), "wgrad")
    assert torch.equal(dW, (dY.double().t() @ X.double()).float())


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attn_ref(qkv, Bf, S, H, dh):
    D = H * dh
    q, k, v = [t.view(Bf, S, H, dh).transpose(1, 2) for t in qkv.view(Bf, S, 3, D).unbind(2)]
    s = (q @ k.transpose(2, 3)) / math.sqrt(dh)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(Bf * S, D)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("S,H,dh,Bf", [(5, 8, 16, 3), (17, 2, 32, 2), (65, 8, 16, 4), (129, 8, 16, 2), (65, 8, 32, 2),
                                       (197, 3, 64, 3), (33, 2, 64, 2), (1025, 8, 16, 1), (224, 1, 64, 1),
                                       # per-frame kernels (S <= 128 and the frame's images fit LDS): more heads than waves,
                                       # whole 32-row tiles, the largest S they take
                                       (40, 12, 16, 2), (96, 4, 32, 3), (128, 8, 16, 2), (64, 3, 64, 5),
                                       # backward keeps the staged side in LDS in chunks (conv1d embedding, S = 1025)
                                       (1025, 8, 32, 1), (1025, 4, 64, 1), (481, 2, 64, 1), (700, 2, 32, 2),
                                       # the benchmarked batch: more (frame, head) workgroups than the chip holds at once
                                       (197, 3, 64, 256), (145, 4, 32, 150)])
def test_attention_fwd_bwd(L, S, H, dh, Bf):
    N = _N()
    assert L.iq_attn_supported(S, dh) == 1
    D = H * dh
    g = torch.Generator(device="cuda").manual_seed(S * 100 + dh)
    qkv = bf(torch.randn(Bf * S, 3 * D, device=dev(), generator=g))
    out = torch.empty(Bf * S, D, dtype=torch.bfloat16, device=dev())
    lse = torch.empty(Bf, H, S, device=dev())
    N.check(L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), Bf, S, H, dh, stream()), "attn_fwd")
    qr = qkv.double().requires_grad_(True)
    oref, lref = attn_ref(qr, Bf, S, H, dh)
    close_bf16(out, oref.detach(), "attn out", rel=2 ** -6)
    close_f32(lse, lref.detach(), "lse", 2e-3)
    dout = bf(torch.randn(Bf * S, D, device=dev(), generator=g))
    dqkv = torch.zeros_like(qkv)
    N.check(L.iq_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), Bf, S, H,
                          dh, stream()), "attn_bwd")
    oref.backward(dout.double())
    gref = qr.grad
    scale = gref.abs().max().item()
    err = (dqkv.double() - gref).abs()
    # P and dS pass through bf16 before their MFMAs: allow 2% of the tensor scale
    assert err.max().item() <= 0.02 * scale + 1e-6, f"attn bwd max err {err.max().item():.4g} scale {scale:.4g}"
    rel_l2 = (err.pow(2).sum() / gref.pow(2).sum()).sqrt().item()
    assert rel_l2 < 8e-3, rel_l2


def test_attention_peaked_softmax_is_stable(L):
    """Large logits: one key dominates each row (online max handling, no inf/nan)."""
    N = _N()
    Bf, S, H, dh = 2, 70, 2, 64
    D = H * dh
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = bf(torch.randn(Bf * S, 3 * D, device=dev(), generator=g) * 6)
    out = torch.empty(Bf * S, D, dtype=torch.bfloat16, device=dev())
    lse = torch.empty(Bf, H, S, device=dev())
    N.check(L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), Bf, S, H, dh, stream()), "attn_fwd")
    oref, lref = attn_ref(qkv.double(), Bf, S, H, dh)
    assert torch.isfinite(out.float()).all()
    close_bf16(out, oref, "peaked", rel=2 ** -6, abs_=0.05)
    close_f32(lse, lref, "lse", 2e-3)


@pytest.mark.parametrize("S,H,dh", [(197, 3, 64), (65, 8, 16), (300, 2, 32)])
def test_attention_deferred_rescale_branch(L, S, H, dh):
    """The forward raises its running maximum (and rescales the output accumulators) only when a key block's maximum
    exceeds it by more than 2^8: a data-dependent, rarely taken branch.  Force it late: one key of the LAST block
    matches one query row so strongly that the maximum jumps by far more than the threshold there -- and, in another
    frame, a spike in the FIRST block that no later block exceeds (no raise after block 0)."""
    N = _N()
    Bf, D = 3, H * dh
    g = torch.Generator(device="cuda").manual_seed(S + dh)
    qkv = torch.randn(Bf * S, 3 * D, device=dev(), generator=g)
    qkv[0 * S + 5, D + 0 * dh: D + 1 * dh] *= 0                                  # frame 0, head 0: key S-3 := 4 q_5
    qkv[0 * S + (S - 3), D: D + dh] = 4.0 * qkv[0 * S + 5, 0:dh]
    qkv[1 * S + 2, D: D + dh] = 4.0 * qkv[1 * S + 40, 0:dh]                       # frame 1, head 0: key 2 := 4 q_40
    qkv = bf(qkv)
    out = torch.empty(Bf * S, D, dtype=torch.bfloat16, device=dev())
    lse = torch.empty(Bf, H, S, device=dev())
    N.check(L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), Bf, S, H, dh, stream()), "attn_fwd")
    oref, lref = attn_ref(qkv.double(), Bf, S, H, dh)
    assert torch.isfinite(out.float()).all()
    close_bf16(out, oref, "deferred rescale", rel=2 ** -6, abs_=0.03)
    close_f32(lse, lref, "lse", 2e-3)
    # the spiked rows really are peaked (the branch was exercised): their output is essentially one value row
    v_row = qkv[0 * S + (S - 3), 2 * D: 2 * D + dh].float()
    assert (out[0 * S + 5, 0:dh].float() - v_row).abs().max().item() < 0.05 * v_row.abs().max().item() + 0.02


def test_attention_limits(L):
    assert L.iq_attn_supported(197, 64) == 1
    assert L.iq_attn_supported(1025, 16) == 1
    assert L.iq_attn_supported(1025, 32) == 1 and L.iq_attn_supported(1025, 64) == 1    # chunked backward
    assert L.iq_attn_supported(4097, 64) == 0   # lse / delta of the whole sequence stay in LDS: documented limit
    assert L.iq_attn_supported(64, 48) == 0


@pytest.mark.parametrize("S,H,dh,Bf,per_head", [(65, 8, 16, 2, False), (197, 3, 64, 2, True), (40, 2, 32, 3, True),
                                                 (600, 2, 64, 1, False)])
def test_attention_mask_branch(L, S, H, dh, Bf, per_head):
    """scale_dot_product_attention.py:30-31: score.masked_fill(mask == 0, -10000) after the 1/sqrt(dh) scaling -- no
    reference caller passes a mask, the branch is part of the layer's surface.  Forward, log-sum-exp and backward
    (masked_fill passes no gradient) against fp64 torch; one query row is fully masked (uniform softmax)."""
    N = _N()
    D = H * dh
    g = torch.Generator(device="cuda").manual_seed(S * 7 + dh)
    qkv = bf(torch.randn(Bf * S, 3 * D, device=dev(), generator=g))
    mask = (torch.rand(Bf, H if per_head else 1, S, S, device=dev(), generator=g) > 0.3)
    mask[:, :, 3, :] = False                       # a fully masked row
    mk = mask.to(torch.uint8).contiguous()
    hs = S * S if per_head and H > 1 else 0
    out = torch.empty(Bf * S, D, dtype=torch.bfloat16, device=dev())
    lse = torch.empty(Bf, H, S, device=dev())
    N.check(L.iq_attn_fwd_masked(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), mk.data_ptr(), hs, Bf, S, H, dh,
                                 stream()), "attn_fwd_masked")
    qr = qkv.double().requires_grad_(True)
    q, k, v = [t.view(Bf, S, H, dh).transpose(1, 2) for t in qr.view(Bf, S, 3, D).unbind(2)]
    sc = (q @ k.transpose(2, 3)) / math.sqrt(dh)
    sc = sc.masked_fill(mask == 0, -10000)
    oref = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(Bf * S, D)
    close_bf16(out, oref.detach(), "masked attn out", rel=2 ** -6)
    close_f32(lse, torch.logsumexp(sc, dim=-1).detach(), "masked lse", 2e-3)
    dout = bf(torch.randn(Bf * S, D, device=dev(), generator=g))
    dqkv = torch.zeros_like(qkv)
    N.check(L.iq_attn_bwd_masked(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                 mk.data_ptr(), hs, Bf, S, H, dh, stream()), "attn_bwd_masked")
    oref.backward(dout.double())
    gref = qr.grad
    scale = gref.abs().max().item()
    err = (dqkv.double() - gref).abs()
    assert err.max().item() <= 0.02 * scale + 1e-6, f"masked attn bwd max err {err.max().item():.4g} scale {scale:.4g}"
    assert (err.pow(2).sum() / gref.pow(2).sum()).sqrt().item() < 8e-3


def test_attention_all_scores_very_negative_stays_finite(L):
    """Rows whose scores are all << 0 have a very negative LSE: exp2(0 - lse) on a padding key overflows fp32.  The
    backward masks padding keys (and only those), so nothing non-finite may reach dQ / dK / dV."""
    N = _N()
    S, H, dh, Bf = 197, 3, 64, 2
    D = H * dh
    qkv = torch.empty(Bf * S, 3 * D, device=dev())
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv[:, :D] = 6.0 + 0.1 * torch.randn(Bf * S, D, device=dev(), generator=g)          # q . k = -64 * 36 / 8 = -288
    qkv[:, D:2 * D] = -6.0 + 0.1 * torch.randn(Bf * S, D, device=dev(), generator=g)
    qkv[:, 2 * D:] = torch.randn(Bf * S, D, device=dev(), generator=g)
    qkv = bf(qkv)
    out = torch.empty(Bf * S, D, dtype=torch.bfloat16, device=dev()); lse = torch.empty(Bf, H, S, device=dev())
    N.check(L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), Bf, S, H, dh, stream()), "attn_fwd")
    assert lse.max().item() < -150
    dout = bf(torch.randn(Bf * S, D, device=dev(), generator=g)); dqkv = torch.empty_like(qkv)
    N.check(L.iq_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), Bf, S, H, dh,
                          stream()), "attn_bwd")
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    qr = qkv.double().requires_grad_(True)
    oref, _ = attn_ref(qr, Bf, S, H, dh)
    oref.backward(dout.double())
    close_bf16(out, oref.detach(), "out", rel=2 ** -6)
    close_bf16(dqkv, qr.grad, "dqkv", rel=2 ** -5, abs_=2e-2 * qr.grad.abs().max().item())


# ------------------------------------------------------------------------------------------------
# embedding front end, head, loss, optimizer
# ------------------------------------------------------------------------------------------------
def test_patchify_matches_conv_unfold(L):
    N = _N()
    Bf, Cc, H, W, p = 3, 2, 16, 24, 4
    x = torch.randn(Bf, Cc, H, W, device=dev())
    P = Cc * p * p
    Kpad = 32
    out = torch.empty(Bf * (H // p) * (W // p), Kpad, dtype=torch.bfloat16, device=dev())
    N.check(L.iq_patchify(x.data_ptr(), out.data_ptr(), 0, Bf, Cc, H, W, p, Kpad, stream()), "patchify")
    ref = torch.nn.functional.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, P)
    assert torch.equal(out[:, :P], bf(ref))
    assert Kpad == P or torch.count_nonzero(out[:, P:]) == 0
    # 1-D
    Lq, k = 64, 8
    x1 = torch.randn(Bf, 2, Lq, device=dev())
    out1 = torch.empty(Bf * (Lq // k), 32, dtype=torch.bfloat16, device=dev())
    N.check(L.iq_patchify(x1.data_ptr(), out1.data_ptr(), 1, Bf, 2, Lq, 0, k, 32, stream()), "patchify1d")
    ref1 = x1.view(Bf, 2, Lq // k, k).permute(0, 2, 1, 3).reshape(-1, 2 * k)
    assert torch.equal(out1[:, :2 * k], bf(ref1))
    assert torch.count_nonzero(out1[:, 2 * k:]) == 0


@pytest.mark.parametrize("pool,with_ln", [(0, False), (0, True), (1, True)])
def test_head_ce_fwd_bwd(L, pool, with_ln):
    N = _N()
    Bf, S, D, K = 37, 9, 192, 19
    g = torch.Generator(device="cuda").manual_seed(11 + pool)
    x = bf(torch.randn(Bf * S, D, device=dev(), generator=g))
    W = torch.randn(K, D, device=dev(), generator=g) / math.sqrt(D)
    b = torch.randn(K, device=dev(), generator=g)
    lg = torch.randn(D, device=dev(), generator=g) if with_ln else None
    lb = torch.randn(D, device=dev(), generator=g) if with_ln else None
    y = torch.randint(0, K, (Bf,), device=dev(), generator=g)
    feat = torch.empty(Bf, D, device=dev())
    hstat = torch.empty(Bf, 2, device=dev())
    logits = torch.empty(Bf, K, device=dev())
    N.check(L.iq_head_fwd(x.data_ptr(), N.ptr(lg), N.ptr(lb), W.data_ptr(), b.data_ptr(), feat.data_ptr(),
                          hstat.data_ptr(), logits.data_ptr(), Bf, S, D, K, pool, stream()), "head_fwd")
    xr = x.double().view(Bf, S, D).requires_grad_(True)
    Wr, br = W.double().requires_grad_(True), b.double().requires_grad_(True)
    f = xr[:, 0] if pool == 0 else xr.mean(1)
    if with_ln:
        lgr, lbr = lg.double().requires_grad_(True), lb.double().requires_grad_(True)
        f = torch.nn.functional.layer_norm(f, (D,), lgr, lbr, 1e-5)
    ref_logits = f @ Wr.t() + br
    close_f32(logits, ref_logits.detach(), "logits", 1e-4)
    loss_ref = torch.nn.functional.cross_entropy(ref_logits, y, label_smoothing=0.1)
    loss_ref.backward()
    loss_sum = torch.zeros(1, device=dev())
    ncorr = torch.zeros(1, dtype=torch.int32, device=dev())
    dlogits = torch.empty(Bf, K, device=dev())
    N.check(L.iq_ce_fwd_bwd(logits.data_ptr(), y.data_ptr(), Bf, K, 0.1, float(Bf), loss_sum.data_ptr(),
                            ncorr.data_ptr(), dlogits.data_ptr(), stream()), "ce")
    assert abs(loss_sum.item() / Bf - loss_ref.item()) < 1e-4
    assert ncorr.item() == int((ref_logits.argmax(1) == y).sum())
    dW, db = torch.empty_like(W), torch.empty_like(b)
    dlg = torch.empty(D, device=dev()) if with_ln else None
    dlb = torch.empty(D, device=dev()) if with_ln else None
    dx = torch.empty_like(x)
    N.check(L.iq_head_bwd(dlogits.data_ptr(), feat.data_ptr(), hstat.data_ptr(), N.ptr(lg), N.ptr(lb), W.data_ptr(),
                          dW.data_ptr(), db.data_ptr(), N.ptr(dlg), N.ptr(dlb), dx.data_ptr(), Bf, S, D, K, pool, 0,
                          stream()), "head_bwd")
    close_f32(dW, Wr.grad, "dW", 1e-3)
    close_f32(db, br.grad, "db", 1e-3)
    close_bf16(dx.view(Bf, S, D), xr.grad, "dx")
    if with_ln:
        close_f32(dlg, lgr.grad, "dln_g", 1e-3)
        close_f32(dlb, lbr.grad, "dln_b", 1e-3)


def test_gradnorm_clip_adamw_match_torch(L):
    N = _N()
    n = 100_000
    g = torch.Generator(device="cuda").manual_seed(2)
    p0 = torch.randn(n, device=dev(), generator=g)
    grad = torch.randn(n, device=dev(), generator=g) * 0.05
    # torch reference: clip_grad_norm_(1.0) + AdamW(lr 1e-4, wd 1e-3, betas (.9,.99)), three steps
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.99))
    p = p0.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=dev())
    ws = torch.empty(L.iq_gradnorm_ws_bytes(n), dtype=torch.uint8, device=dev())
    gn = torch.empty(1, device=dev())
    for step in (1, 2, 3):
        gstep = grad * step
        pr.grad = gstep.clone()
        total = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        N.check(L.iq_gradnorm_sq(gstep.data_ptr(), n, 1.0, ws.data_ptr(), gn.data_ptr(), stream()), "gradnorm")
        assert abs(math.sqrt(gn.item()) - total.item()) < 1e-4 * total.item()
        N.check(L.iq_adamw_step(p.data_ptr(), gstep.data_ptr(), m.data_ptr(), v.data_ptr(), shadow.data_ptr(), n, 1e-4,
                                0.9, 0.99, 1e-8, 1e-3, step, gn.data_ptr(), 1.0, 1.0, None, stream()), "adamw")
        assert torch.allclose(p, pr.data, atol=2e-7, rtol=1e-6), (p - pr.data).abs().max()
        assert torch.equal(shadow, p.to(torch.bfloat16))
    # dyn (device lr/step) path gives the same update as host scalars
    p2, m2, v2 = p0.clone(), torch.zeros_like(p), torch.zeros_like(p)
    p3, m3, v3 = p0.clone(), torch.zeros_like(p), torch.zeros_like(p)
    dyn = torch.tensor([3e-4, 1.0], device=dev())
    N.check(L.iq_adamw_step(p2.data_ptr(), grad.data_ptr(), m2.data_ptr(), v2.data_ptr(), None, n, 0.0, 0.9, 0.99, 1e-8,
                            1e-3, 0, None, 0.0, 1.0, dyn.data_ptr(), stream()), "adamw dyn")
    N.check(L.iq_adamw_step(p3.data_ptr(), grad.data_ptr(), m3.data_ptr(), v3.data_ptr(), None, n, 3e-4, 0.9, 0.99, 1e-8,
                            1e-3, 1, None, 0.0, 1.0, None, stream()), "adamw host")
    assert torch.equal(p2, p3)


def test_cast_and_transpose(L):
    N = _N()
    src = torch.randn(300, 200, device=dev())
    dst = torch.empty(200, 300, dtype=torch.bfloat16, device=dev())
    N.check(L.iq_transpose_cast_bf16(src.data_ptr(), dst.data_ptr(), 300, 200, stream()), "transpose")
    assert torch.equal(dst, bf(src.t().contiguous()))
    flat = torch.randn(1000, device=dev())
    out = torch.empty(1000, dtype=torch.bfloat16, device=dev())
    N.check(L.iq_cast_bf16(flat.data_ptr(), out.data_ptr(), 1000, stream()), "cast")
    assert torch.equal(out, bf(flat))
