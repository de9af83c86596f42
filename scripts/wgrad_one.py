"""Run ONE wgrad configuration a few times (for rocprofv3 --pmc passes). usage: python3 scripts/wgrad_one.py N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
N_, K = int(sys.argv[1]), int(sys.argv[2]); M = 50432
dY = torch.randn(M, N_, device=d).bfloat16(); X = torch.randn(M, K, device=d).bfloat16()
dW = torch.empty(N_, K, device=d); db = torch.empty(N_, device=d)
nb = L.iq_wgrad_ws_bytes(M, N_, K); ws = torch.empty(nb, dtype=torch.uint8, device=d)
st = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N_, K, ws.data_ptr(), nb, 0, st)
torch.cuda.synchronize()
