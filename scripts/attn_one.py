"""One attention forward + backward pair of cfg B (256 frames x 3 heads, S = 197, dh = 64), a few repetitions: the target
of scripts/pmc_any.sh (usage: scripts/pmc_any.sh <tag> <kernel substring> scripts/attn_one.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
S, H, dh, B = 197, 3, 64, 256; D = H * dh
st = torch.cuda.current_stream().cuda_stream
qkv = torch.randn(B * S, 3 * D, device=d).bfloat16(); out = torch.empty(B * S, D, device=d, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=d); dout = torch.randn(B * S, D, device=d).bfloat16(); dqkv = torch.empty_like(qkv)
for _ in range(6):
    L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, dh, st)
    L.iq_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, S, H, dh, st)
torch.cuda.synchronize()
