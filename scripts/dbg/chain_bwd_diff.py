"""iq_ffn_chain_bwd against the two launches it replaces (gate GEMM, then GEMM + LayerNorm backward): differences + timing.
   python scripts/dbg/chain_bwd_diff.py frames S D F pdrop"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
frames, S, D, F = (int(v) for v in sys.argv[1:5]); pdrop = float(sys.argv[5])
M = frames * S
st = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(1)
bf = lambda t: t.to(torch.bfloat16)
dO = bf(torch.randn(M, D, device=d, generator=g)); W2t = bf(torch.randn(F, D, device=d, generator=g) / math.sqrt(D)); W1t = bf(torch.randn(D, F, device=d, generator=g) / math.sqrt(F))
R = bf(torch.randn(M, D, device=d, generator=g)); z = bf(torch.randn(M, D, device=d, generator=g) * 1.5 + 0.3)
gm = torch.rand(D, device=d, generator=g) + 0.5
# the forward kernel produces H and its gate bits
X1 = bf(torch.randn(M, D, device=d, generator=g)); W1 = bf(torch.randn(F, D, device=d, generator=g) / math.sqrt(D)); W2 = bf(torch.randn(D, F, device=d, generator=g) / math.sqrt(F))
b1 = torch.randn(F, device=d, generator=g); b2 = torch.randn(D, device=d, generator=g); bt = torch.randn(D, device=d, generator=g)
hid = torch.empty(M, F, device=d, dtype=torch.bfloat16); Zf = torch.empty(M, D, device=d, dtype=torch.bfloat16); Xf = torch.empty_like(Zf)
mf = torch.empty(M, device=d); rf = torch.empty(M, device=d)
gate = torch.zeros(L.iq_ffn_chain_gate_bytes(M, F), dtype=torch.uint8, device=d)
xd = N.Dropout(); xd.p = pdrop; xd.seed = 3; xd.site = 2; xd.step = 1
N.check(L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(xd) if pdrop > 0 else None, hid.data_ptr(), W2.data_ptr(), b2.data_ptr(), None,
                           gm.data_ptr(), bt.data_ptr(), 1e-12, Zf.data_ptr(), Xf.data_ptr(), mf.data_ptr(), rf.data_ptr(), gate.data_ptr(), frames, S, D, F, st()), "fwd")
mean = z.float().mean(-1).contiguous(); rstd = (1 / torch.sqrt(z.float().var(-1, unbiased=False) + 1e-12)).contiguous()
x = N.Dropout(); x.p = pdrop; x.seed = 77; x.site = 5; x.step = 3
dr = C.byref(x) if pdrop > 0 else None
scale = 1.0 / (1 - pdrop) if pdrop > 0 else 1.0
gH0 = torch.empty(M, F, device=d, dtype=torch.bfloat16); dz0 = torch.empty_like(z); dy0 = torch.zeros_like(z)
rows0 = L.iq_gemm_lnbwd_partial_rows(M); part0 = torch.empty(rows0, 2 * D, device=d)
e = N.Epilogue(); e.gate = hid.data_ptr(); e.ldg = F; e.gate_scale = scale
def two():
    N.check(L.iq_gemm_bf16_nt(dO.data_ptr(), D, W2t.data_ptr(), D, gH0.data_ptr(), F, M, F, D, C.byref(e), st()), "gate")
    N.check(L.iq_gemm_bf16_lnbwd(gH0.data_ptr(), F, W1t.data_ptr(), F, R.data_ptr(), D, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gm.data_ptr(), dr,
                                 dz0.data_ptr(), dy0.data_ptr(), part0.data_ptr(), M, D, F, st()), "lnbwd")
gH1 = torch.zeros(M, F, device=d, dtype=torch.bfloat16); dz1 = torch.zeros_like(z); dy1 = torch.zeros_like(z)
rows1 = L.iq_ffn_chain_bwd_partial_rows(M); part1 = torch.zeros(rows1, 2 * D, device=d)
def one():
    N.check(L.iq_ffn_chain_bwd(dO.data_ptr(), W2t.data_ptr(), gate.data_ptr(), scale, gH1.data_ptr(), W1t.data_ptr(), R.data_ptr(), z.data_ptr(), mean.data_ptr(),
                               rstd.data_ptr(), gm.data_ptr(), dr, dz1.data_ptr(), dy1.data_ptr(), part1.data_ptr(), None, None, frames, S, D, F, st()), "chain_bwd")
Wot = (torch.randn(D, D, device=d) / D ** 0.5).bfloat16(); dA = torch.zeros_like(z); dA0 = torch.zeros_like(z)
def one_dA():
    N.check(L.iq_ffn_chain_bwd(dO.data_ptr(), W2t.data_ptr(), gate.data_ptr(), scale, gH1.data_ptr(), W1t.data_ptr(), R.data_ptr(), z.data_ptr(), mean.data_ptr(),
                               rstd.data_ptr(), gm.data_ptr(), dr, dz1.data_ptr(), dy1.data_ptr(), part1.data_ptr(), Wot.data_ptr(), dA.data_ptr(), frames, S, D, F, st()), "chain_bwd")
def outproj():
    N.check(L.iq_gemm_bf16_nt((dy1 if dr is not None else dz1).data_ptr(), D, Wot.data_ptr(), D, dA0.data_ptr(), D, M, D, D, None, st()), "out-proj dgrad")
two(); one(); torch.cuda.synchronize()
for name, a, b in (("gH", gH0, gH1), ("dZ", dz0, dz1), ("dY", dy0, dy1)):
    ne = a.view(torch.int16) != b.view(torch.int16)
    print(f"{name}: {ne.float().mean().item():.5f} of elements differ; max abs diff {(a.float() - b.float()).abs().max().item():.4g} (scale {a.float().abs().max().item():.3g})")
s0, s1 = part0.sum(0), part1.sum(0)
print(f"dgamma|dbeta: max abs diff {(s0 - s1).abs().max().item():.4g} (scale {s0.abs().max().item():.4g})")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print(f"two launches {timeit(two):.1f} us, chain {timeit(one):.1f} us; out-proj dgrad {timeit(outproj):.1f} us, chain with it {timeit(one_dA):.1f} us")
