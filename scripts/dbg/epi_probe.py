import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
M = N_ = K = 64
A = torch.eye(64, device=d).bfloat16()
for name, B in (("col", torch.arange(64, device=d).float().view(64, 1).expand(64, 64)), ("row", torch.arange(64, device=d).float().view(1, 64).expand(64, 64))):
    B = B.contiguous().bfloat16()
    out = torch.zeros(M, N_, device=d, dtype=torch.bfloat16)
    L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, out.data_ptr(), N_, M, N_, K, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.float().cpu()
    print(name, "row0:", o[0].int().tolist())
    print(name, "col0:", o[:, 0].int().tolist())
