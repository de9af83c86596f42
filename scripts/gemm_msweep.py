"""Time the NT GEMM as a function of M (row tiles) to expose workgroup-slot quantisation (staircase vs linear).
usage: python scripts/gemm_msweep.py"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib()
d = torch.device("cuda:0")
reps = 30
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
def nt(M, N_, K, res=False):
    A = torch.randn(M, K, device=d).bfloat16(); B = (torch.randn(N_, K, device=d) / math.sqrt(K)).bfloat16()
    Cc = torch.empty(M, N_, device=d, dtype=torch.bfloat16)
    e = N.Epilogue(); keep = []
    if res: t = torch.randn(M, N_, device=d).bfloat16(); keep.append(t); e.residual = t.data_ptr(); e.ldr = N_
    return timeit(lambda: L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), st()))
for (N_, K, res) in [(192, 192, True), (768, 192, False), (192, 768, True), (576, 192, False)]:
    print(f"--- N={N_} K={K} res={int(res)}")
    for tm in list(range(32, 513, 32)) + [394] + list(range(576, 1025, 64)):
        M = tm * 128
        us = nt(M, N_, K, res)
        byt = 2 * (M * K + N_ * K + M * N_) + (2 * M * N_ if res else 0)
        print(f"tiles_m={tm:5d} M={M:7d}: {us:7.1f} us  {byt/us/1e3:7.1f} GB/s  us/tile_m={us/tm*1e3:6.1f} ns", flush=True)
