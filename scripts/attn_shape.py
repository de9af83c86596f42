"""Attention forward / backward time at one shape:  python scripts/attn_shape.py S H dh B   (HIP-event timing, 20 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
S, H, dh, B = (int(v) for v in sys.argv[1:5])
D = H * dh
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
st = lambda: torch.cuda.current_stream().cuda_stream
qkv = torch.randn(B * S, 3 * D, device=d).bfloat16(); out = torch.empty(B * S, D, device=d, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=d); dout = torch.randn(B * S, D, device=d).bfloat16(); dqkv = torch.empty_like(qkv)
tf = timeit(lambda: L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, dh, st()))
tb = timeit(lambda: L.iq_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, S, H, dh, st()))
bf, bb = 2 * B * S * 4 * D, 2 * B * S * 8 * D
print(f"S={S} H={H} dh={dh} B={B} frame={os.environ.get('IQ_TUNE_ATTN_FRAME', 'auto')}: fwd {tf:6.1f} us ({bf / tf / 1e6:6.2f} TB/s)   bwd {tb:6.1f} us ({bb / tb / 1e6:6.2f} TB/s)")
