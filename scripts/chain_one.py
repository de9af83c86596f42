"""The two one-launch layer kernels of cfg B (256 frames x 197 tokens, D 192, F 768, dropout 0.1), a few repetitions: the target of
scripts/pmc_any.sh (usage: scripts/pmc_any.sh <tag> ffn_chain_fwd|ffn_chain_bwd scripts/chain_one.py)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
B, S, D, F = 256, 197, 192, 768
M = B * S
st = torch.cuda.current_stream().cuda_stream
bf = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).bfloat16()
def dr(site):
    x = N.Dropout(); x.p = 0.1; x.seed = 1; x.site = site; x.step = 3
    return x
A, R = bf(M, D), bf(M, D)
Wo, W1, W2, Wq = bf(D, D, sc=1 / math.sqrt(D)), bf(F, D, sc=1 / math.sqrt(D)), bf(D, F, sc=1 / math.sqrt(F)), bf(3 * D, D, sc=1 / math.sqrt(D))
bo, b1, b2, bq = (torch.randn(n, device=d) for n in (D, F, D, 3 * D))
g1, g2 = (torch.rand(D, device=d) + 0.5 for _ in range(2)); be1, be2 = (torch.randn(D, device=d) for _ in range(2))
new = lambda *s, dt=torch.bfloat16: torch.empty(*s, dtype=dt, device=d)
Z1, X1, H, Z2, X, Yq = new(M, D), new(M, D), new(M, F), new(M, D), new(M, D), new(M, 3 * D)
m1, r1, m2, r2 = (new(M, dt=torch.float32) for _ in range(4))
gate = torch.zeros(L.iq_ffn_chain_gate_bytes(M, F), dtype=torch.uint8, device=d)
d0, d1, d2 = dr(1), dr(2), dr(3)
dO, gH, dz, dy, dA = bf(M, D), new(M, F), new(M, D), new(M, D), new(M, D)
part = new(L.iq_ffn_chain_bwd_partial_rows(M), 2 * D, dt=torch.float32)
W2t, W1t, Wot = W2.t().contiguous(), W1.t().contiguous(), Wo.t().contiguous()
for _ in range(6):
    N.check(L.iq_attn_out_ffn_chain_fwd(A.data_ptr(), Wo.data_ptr(), bo.data_ptr(), C.byref(d0), R.data_ptr(), g1.data_ptr(), be1.data_ptr(),
                                        Z1.data_ptr(), X1.data_ptr(), m1.data_ptr(), r1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(d1),
                                        H.data_ptr(), W2.data_ptr(), b2.data_ptr(), C.byref(d2), g2.data_ptr(), be2.data_ptr(), 1e-12,
                                        Z2.data_ptr(), X.data_ptr(), m2.data_ptr(), r2.data_ptr(), gate.data_ptr(), Wq.data_ptr(), bq.data_ptr(),
                                        Yq.data_ptr(), B, S, D, F, st), "fwd")
    N.check(L.iq_ffn_chain_bwd(dO.data_ptr(), W2t.data_ptr(), gate.data_ptr(), 1.0 / 0.9, gH.data_ptr(), W1t.data_ptr(), R.data_ptr(), Z1.data_ptr(),
                               m1.data_ptr(), r1.data_ptr(), g1.data_ptr(), C.byref(d0), dz.data_ptr(), dy.data_ptr(), part.data_ptr(),
                               Wot.data_ptr(), dA.data_ptr(), B, S, D, F, st), "bwd")
torch.cuda.synchronize()
