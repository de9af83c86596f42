"""Run the grouped weight-gradient launch of one ViT-Tiny encoder layer (cfg B: M = 50432) a few times and time it.
usage: python3 scripts/wgrad_group_one.py [reps]   (also the target of scripts/pmc_any.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
M = 50432; D, F = 192, 768
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shapes = [(D, F), (F, D), (D, D), (3 * D, D)]
probs = (N.WgradProblem * 4)(); keep = []
byt = 0
for i, (n, k) in enumerate(shapes):
    dY = torch.randn(M, n, device=d).bfloat16(); X = torch.randn(M, k, device=d).bfloat16()
    dW = torch.empty(n, k, device=d); db = torch.empty(n, device=d); keep += [dY, X, dW, db]
    probs[i].dY = dY.data_ptr(); probs[i].ldy = n; probs[i].X = X.data_ptr(); probs[i].ldx = k
    probs[i].dW = dW.data_ptr(); probs[i].dbias = db.data_ptr(); probs[i].N = n; probs[i].K = k
    byt += 2 * M * (n + k) + 4 * n * k
nb = L.iq_wgrad_grouped_ws_bytes(probs, 4, M, budget); ws = torch.empty(nb, dtype=torch.uint8, device=d)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.iq_gemm_bf16_wgrad_grouped(probs, 4, M, ws.data_ptr(), nb, 0, budget, None, 0, st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    L.iq_gemm_bf16_wgrad_grouped(probs, 4, M, ws.data_ptr(), nb, 0, budget, None, 0, st)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / reps * 1e3
print(f"grouped layer wgrad: {us:.1f} us  algorithmic {byt / 1e6:.1f} MB -> {byt / us / 1e3:.0f} GB/s  slab ws {nb / 1e6:.1f} MB")
