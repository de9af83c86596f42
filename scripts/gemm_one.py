"""Run ONE NT-GEMM configuration a few times (for rocprofv3 --pmc passes).
usage: python3 scripts/gemm_one.py N K [bias relu drop res gate]"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
N_, K = int(sys.argv[1]), int(sys.argv[2]); flags = sys.argv[3:]
M = 50432
A = torch.randn(M, K, device=d).bfloat16(); B = (torch.randn(N_, K, device=d) / math.sqrt(K)).bfloat16()
Cc = torch.empty(M, N_, device=d, dtype=torch.bfloat16)
e = N.Epilogue(); keep = []
if "bias" in flags: t = torch.randn(N_, device=d); keep.append(t); e.bias = t.data_ptr()
if "relu" in flags: e.relu = 1
if "drop" in flags: e.drop.p = 0.1; e.drop.seed = 1
if "res" in flags: t = torch.randn(M, N_, device=d).bfloat16(); keep.append(t); e.residual = t.data_ptr(); e.ldr = N_
st = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), st)
torch.cuda.synchronize()
