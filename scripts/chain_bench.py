"""Time iq_gemm_bf16_chain against the two iq_gemm_bf16_nt calls it replaces (cfg B FFN shapes, M = 50432)."""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
M, D, F = 50432, 192, 768
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
X = torch.randn(M, D, device=d).bfloat16(); Wa = (torch.randn(F, D, device=d) / math.sqrt(D)).bfloat16()
Wb = (torch.randn(D, F, device=d) / math.sqrt(F)).bfloat16()
ba, bb = torch.randn(F, device=d), torch.randn(D, device=d)
R = torch.randn(M, D, device=d).bfloat16(); G = torch.randn(M, F, device=d).bfloat16()
H = torch.empty(M, F, device=d, dtype=torch.bfloat16); Y = torch.empty(M, D, device=d, dtype=torch.bfloat16)
def epi(**kw):
    e = N.Epilogue()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor): setattr(e, k, v.data_ptr())
        elif k == "p": e.drop.p = v; e.drop.seed = 1; e.drop.site = kw.get("site", 2)
        elif k != "site": setattr(e, k, v)
    return e
cases = {"forward p=0.1": (epi(bias=ba, relu=1, p=0.1, site=2), epi(bias=bb, p=0.1, site=3, residual=R, ldr=D)),
         "forward eval": (epi(bias=ba, relu=1), epi(bias=bb, residual=R, ldr=D)),
         "data-grad chain": (epi(gate=G, ldg=F, gate_scale=1.11), epi(residual=R, ldr=D))}
for name, (e1, e2) in cases.items():
    t2 = timeit(lambda: (L.iq_gemm_bf16_nt(X.data_ptr(), D, Wa.data_ptr(), D, H.data_ptr(), F, M, F, D, C.byref(e1), st()),
                         L.iq_gemm_bf16_nt(H.data_ptr(), F, Wb.data_ptr(), F, Y.data_ptr(), D, M, D, F, C.byref(e2), st())))
    t1 = timeit(lambda: L.iq_gemm_bf16_chain(X.data_ptr(), D, Wa.data_ptr(), D, H.data_ptr(), F, Wb.data_ptr(), F, Y.data_ptr(), D,
                                             M, F, D, C.byref(e1), C.byref(e2), st()))
    print(f"{name:18s}: two GEMMs {t2:6.1f} us   chained {t1:6.1f} us")
