"""Time iq_ffn_chain_fwd (cfg B shape) against ablated builds of ffn_chain.hip (timing only; results wrong):
   python scripts/dbg/chain_ablate.py <lib path> [pdrop]"""
import ctypes as C, math, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import vit_vs_raw_iq_amd._native as N
if sys.argv[1] != "-":
    N.LIB_PATH = os.path.join(root, sys.argv[1])
import torch
L = N.lib(); d = torch.device("cuda:0")
pdrop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
frames, S, D, F = 256, 197, 192, 768
M = frames * S
st = lambda: torch.cuda.current_stream().cuda_stream
bf = lambda *s: torch.randn(*s, device=d).bfloat16()
X1 = bf(M, D); W1 = (torch.randn(F, D, device=d) / math.sqrt(D)).bfloat16(); W2 = (torch.randn(D, F, device=d) / math.sqrt(F)).bfloat16()
b1 = torch.randn(F, device=d); b2 = torch.randn(D, device=d); gm = torch.rand(D, device=d) + 0.5; bt = torch.randn(D, device=d)
H = torch.empty(M, F, device=d, dtype=torch.bfloat16); Z = torch.empty(M, D, device=d, dtype=torch.bfloat16); X = torch.empty_like(Z)
mean = torch.empty(M, device=d); rstd = torch.empty(M, device=d)
def dr(site):
    x = N.Dropout(); x.p = pdrop; x.seed = 1; x.site = site; x.step = 3
    return x
d1, d2 = dr(2), dr(3)
fn = lambda: L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(d1) if pdrop > 0 else None, H.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                C.byref(d2) if pdrop > 0 else None, gm.data_ptr(), bt.data_ptr(), 1e-12, Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                None, frames, S, D, F, st())
for _ in range(5): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): fn()
b.record(); torch.cuda.synchronize()
print(f"{sys.argv[1]:40s} pdrop {pdrop}: {a.elapsed_time(b) / 30 * 1e3:7.1f} us")
