"""Where does iq_ffn_chain_fwd differ from FFN1 -> FFN2+LN?  python scripts/dbg/chain_diff.py frames S D F pdrop"""
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
X1 = bf(torch.randn(M, D, device=d, generator=g)); W1 = bf(torch.randn(F, D, device=d, generator=g) / math.sqrt(D)); W2 = bf(torch.randn(D, F, device=d, generator=g) / math.sqrt(F))
b1 = torch.randn(F, device=d, generator=g); b2 = torch.randn(D, device=d, generator=g); gm = torch.rand(D, device=d, generator=g) + 0.5; bt = torch.randn(D, device=d, generator=g)
def drop(site):
    x = N.Dropout(); x.p = pdrop; x.seed = 77; x.site = site; x.step = 3
    return x
d1, d2 = drop(5), drop(6)
e = N.Epilogue(); e.bias = b1.data_ptr(); e.relu = 1
if pdrop > 0: e.drop = d1
H0 = torch.empty(M, F, device=d, dtype=torch.bfloat16)
N.check(L.iq_gemm_bf16_nt(X1.data_ptr(), D, W1.data_ptr(), D, H0.data_ptr(), F, M, F, D, C.byref(e), st()), "nt")
Z0 = torch.empty(M, D, device=d, dtype=torch.bfloat16); X0 = torch.empty_like(Z0); m0 = torch.empty(M, device=d); r0 = torch.empty(M, device=d)
N.check(L.iq_gemm_bf16_ln(H0.data_ptr(), F, W2.data_ptr(), F, b2.data_ptr(), X1.data_ptr(), D, C.byref(d2) if pdrop > 0 else None, gm.data_ptr(), bt.data_ptr(), 1e-12,
                          Z0.data_ptr(), X0.data_ptr(), m0.data_ptr(), r0.data_ptr(), M, D, F, st()), "ln")
H1 = torch.zeros(M, F, device=d, dtype=torch.bfloat16); Z1 = torch.zeros(M, D, device=d, dtype=torch.bfloat16); X2 = torch.zeros_like(Z1); m1 = torch.zeros(M, device=d); r1 = torch.zeros(M, device=d)
N.check(L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(d1) if pdrop > 0 else None, H1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                           C.byref(d2) if pdrop > 0 else None, gm.data_ptr(), bt.data_ptr(), 1e-12, Z1.data_ptr(), X2.data_ptr(), m1.data_ptr(), r1.data_ptr(), None, frames, S, D, F, st()), "chain")
torch.cuda.synchronize()
for name, a, b in (("H", H0, H1), ("Z", Z0, Z1), ("X", X0, X2)):
    ne = a.view(torch.int16) != b.view(torch.int16)
    print(f"{name}: {ne.float().mean().item():.4f} of elements differ; max abs diff {(a.float() - b.float()).abs().max().item():.4g}")
    if ne.any():
        rows = ne.any(1).nonzero().flatten(); cols = ne.any(0).nonzero().flatten()
        print(f"   rows with a difference: {len(rows)} of {a.shape[0]}, first {rows[:12].tolist()}, row % 32 histogram {torch.bincount(rows % 32, minlength=32).tolist()}")
        print(f"   cols with a difference: {len(cols)} of {a.shape[1]}, first {cols[:16].tolist()}, col % 64 histogram {torch.bincount(cols % 64, minlength=64).tolist()}")
        r = rows[0].item(); cc = ne[r].nonzero().flatten()[:8].tolist()
        print(f"   row {r}: cols {cc}: ref {[round(a[r, c].item(), 3) for c in cc]} got {[round(b[r, c].item(), 3) for c in cc]}")
