"""Attention forward / backward time vs number of (frame, head) workgroups: is the 768-workgroup launch of cfg B
(1.5 rounds over 512 resident slots) paying for two rounds?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
S, H, dh = 197, 3, 64; D = H * dh
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
st = lambda: torch.cuda.current_stream().cuda_stream
for B in (85, 128, 170, 171, 213, 256, 341, 342, 512):
    qkv = torch.randn(B * S, 3 * D, device=d).bfloat16(); out = torch.empty(B * S, D, device=d, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=d); dout = torch.randn(B * S, D, device=d).bfloat16(); dqkv = torch.empty_like(qkv)
    tf = timeit(lambda: L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, dh, st()))
    tb = timeit(lambda: L.iq_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, S, H, dh, st()))
    print(f"B={B:4d} workgroups={B*H:5d} ({B*H/512:4.2f} x 512 slots): fwd {tf:6.1f} us ({tf/(B*H)*1e3:5.1f} ns/WG)   bwd {tb:6.1f} us ({tb/(B*H)*1e3:5.1f} ns/WG)")
