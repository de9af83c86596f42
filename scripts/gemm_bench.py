"""Micro-benchmark of the C-ABI GEMM entry points on the ViT-Tiny / raw-IQ shapes (HIP-event timing).
usage: python scripts/gemm_bench.py [M]"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib()
d = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
reps = 20
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
def nt(N_, K, bias=False, relu=False, drop=0.0, res=False, gate=False):
    A = torch.randn(M, K, device=d).bfloat16(); B = (torch.randn(N_, K, device=d) / math.sqrt(K)).bfloat16()
    Cc = torch.empty(M, N_, device=d, dtype=torch.bfloat16)
    e = N.Epilogue(); keep = []
    if bias: t = torch.randn(N_, device=d); keep.append(t); e.bias = t.data_ptr()
    e.relu = 1 if relu else 0
    if drop > 0: e.drop.p = drop; e.drop.seed = 1; e.drop.site = 2; e.drop.step = 3
    if res: t = torch.randn(M, N_, device=d).bfloat16(); keep.append(t); e.residual = t.data_ptr(); e.ldr = N_
    if gate: t = torch.randn(M, N_, device=d).bfloat16(); keep.append(t); e.gate = t.data_ptr(); e.ldg = N_; e.gate_scale = 1.1
    us = timeit(lambda: L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), st()))
    byt = 2 * (M * K + N_ * K + M * N_) + (2 * M * N_ if res else 0) + (2 * M * N_ if gate else 0)
    print(f"nt   N={N_:5d} K={K:5d} bias={int(bias)} relu={int(relu)} drop={drop} res={int(res)} gate={int(gate)}: {us:8.1f} us  {byt/us/1e3:7.1f} GB/s  {2*M*N_*K/us/1e6:7.1f} TFLOP/s")
def wg(N_, K):
    dY = torch.randn(M, N_, device=d).bfloat16(); X = torch.randn(M, K, device=d).bfloat16()
    dW = torch.empty(N_, K, device=d); db = torch.empty(N_, device=d)
    nb = L.iq_wgrad_ws_bytes(M, N_, K); ws = torch.empty(nb, dtype=torch.uint8, device=d)
    us = timeit(lambda: L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N_, K, ws.data_ptr(), nb, 0, st()))
    byt = 2 * (M * N_ + M * K) + 4 * N_ * K
    print(f"wgrad N={N_:5d} K={K:5d}: {us:8.1f} us  {byt/us/1e3:7.1f} GB/s  {2*M*N_*K/us/1e6:7.1f} TFLOP/s  (slab ws {nb/1e6:.1f} MB)")
print(f"M = {M}")
if len(sys.argv) > 4 and sys.argv[2] == "shape":      # python scripts/gemm_bench.py M shape N K [res|gate|bias ...]
    fl = sys.argv[5:]
    nt(int(sys.argv[3]), int(sys.argv[4]), bias="bias" in fl, relu="relu" in fl, drop=0.1 if "drop" in fl else 0.0, res="res" in fl, gate="gate" in fl)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "big":        # ViT-Base shapes (M = 512 x 197 = 100864)
    for (n_, k_, kw) in ((2304, 768, dict(bias=True)), (768, 768, dict(bias=True, drop=0.1, res=True)), (3072, 768, dict(bias=True, relu=True, drop=0.1)),
                         (768, 3072, dict(bias=True, drop=0.1, res=True)), (3072, 768, dict(gate=True)), (768, 3072, dict(res=True)), (768, 2304, dict(res=True)), (768, 768, dict())):
        nt(n_, k_, **kw)
    wg(768, 768); wg(2304, 768); wg(3072, 768); wg(768, 3072)
    sys.exit(0)
if len(sys.argv) > 2:
    nt = wg = lambda *a, **k: None
nt(576, 192, bias=True)
nt(192, 192)
nt(192, 192, bias=True, res=True)
nt(192, 192, bias=True, drop=0.1, res=True)
nt(768, 192)
nt(768, 192, bias=True, relu=True)
nt(768, 192, bias=True, relu=True, drop=0.1)
nt(192, 768)
nt(192, 768, bias=True, drop=0.1, res=True)
nt(768, 192, gate=True)
nt(192, 576, res=True)
wg(192, 768); wg(768, 192); wg(192, 192); wg(576, 192)
def gemm_ln(D, K, drop=0.1):
    """fused GEMM + dropout + residual + LayerNorm vs the two launches it replaces"""
    A = torch.randn(M, K, device=d).bfloat16(); W = (torch.randn(D, K, device=d) / math.sqrt(K)).bfloat16()
    R = torch.randn(M, D, device=d).bfloat16(); bias = torch.randn(D, device=d); gm = torch.rand(D, device=d) + 0.5; bt = torch.randn(D, device=d)
    Z = torch.empty(M, D, device=d, dtype=torch.bfloat16); X = torch.empty_like(Z); mean = torch.empty(M, device=d); rstd = torch.empty(M, device=d)
    dr = N.Dropout(); dr.p = drop; dr.seed = 1; dr.site = 2; dr.step = 3
    e = N.Epilogue(); e.bias = bias.data_ptr(); e.drop = dr; e.residual = R.data_ptr(); e.ldr = D
    def two():
        L.iq_gemm_bf16_nt(A.data_ptr(), K, W.data_ptr(), K, Z.data_ptr(), D, M, D, K, C.byref(e), st())
        L.iq_ln_fwd(Z.data_ptr(), gm.data_ptr(), bt.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D, 1e-12, st())
    def one():
        L.iq_gemm_bf16_ln(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), R.data_ptr(), D, C.byref(dr), gm.data_ptr(), bt.data_ptr(), 1e-12,
                          Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D, K, st())
    t2, t1 = timeit(two), timeit(one)
    byt = 2 * (M * K + D * K + 3 * M * D) + 8 * M
    print(f"gemm+ln D={D:4d} K={K:5d} drop={drop}: two launches {t2:7.1f} us, fused {t1:7.1f} us  ({byt/t1/1e3:7.1f} GB/s algorithmic)")
if len(sys.argv) > 2 and sys.argv[2] == "ln":
    for D_, K_ in ((192, 192), (192, 768), (128, 128), (128, 1024), (256, 256), (256, 1024)):
        gemm_ln(D_, K_, 0.1); gemm_ln(D_, K_, 0.0)
