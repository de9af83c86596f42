"""Race screen for the ping-pong kernels (gemm_big, wgrad_big, the gemm_ln band variant): every kernel is deterministic, so
any run-to-run difference of its output is a missing wait or barrier.  Each case runs REPS times, alternating with a
different kernel (changes what else is in flight) and compares bit for bit with the first run.
usage: python scripts/dbg/race_screen.py [reps]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib()
d = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def st(): return torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=d).manual_seed(5)
def rn(*s): return torch.randn(*s, device=d, generator=g)
bad = 0
def screen(name, run, outs, perturb):
    global bad
    run(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    for r in range(REPS):
        if r & 1: perturb()
        run()
        if r % 3 == 0: perturb()
        torch.cuda.synchronize()
        for o, q in zip(outs, ref):
            if not torch.equal(o, q):
                bad += 1
                print(f"MISMATCH {name} rep {r}: {(o.float() - q.float()).abs().max().item():.4g}, {(o != q).sum().item()} elements")
                break
    print(f"{name}: {REPS} runs identical" if not bad else f"{name}: done")
# a perturbing kernel: an unrelated small GEMM
pa = rn(8192, 192).bfloat16(); pb = rn(768, 192).bfloat16(); pc = torch.empty(8192, 768, device=d, dtype=torch.bfloat16)
def perturb(): L.iq_gemm_bf16_nt(pa.data_ptr(), 192, pb.data_ptr(), 192, pc.data_ptr(), 768, 8192, 768, 192, None, st())
for (M, N_, K) in ((50432, 768, 768), (100864, 768, 3072), (40077, 1024, 256), (35072, 2304, 768)):
    A = rn(M, K).bfloat16(); B = (rn(N_, K) / math.sqrt(K)).bfloat16(); Cc = torch.empty(M, N_, device=d, dtype=torch.bfloat16)
    R = rn(M, N_).bfloat16(); bias = rn(N_)
    e = N.Epilogue(); e.bias = bias.data_ptr(); e.residual = R.data_ptr(); e.ldr = N_
    e.drop.p = 0.1; e.drop.seed = 3; e.drop.site = 1; e.drop.step = 9
    screen(f"gemm_big {M}x{N_}x{K}", lambda: L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), st()), [Cc], perturb)
    del A, B, Cc, R
for (M, shapes) in ((50432, [(192, 768), (768, 192), (192, 192), (576, 192)]), (16640, [(128, 1024), (1024, 128), (128, 128), (384, 128)]),
                    (25216, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])):
    probs = (N.WgradProblem * len(shapes))(); keep = []; outs = []
    for i, (n, k) in enumerate(shapes):
        dY = rn(M, n).bfloat16(); X = rn(M, k).bfloat16(); dW = torch.zeros(n, k, device=d); db = torch.zeros(n, device=d)
        keep += [dY, X]; outs += [dW, db]
        probs[i].dY = dY.data_ptr(); probs[i].ldy = n; probs[i].X = X.data_ptr(); probs[i].ldx = k
        probs[i].dW = dW.data_ptr(); probs[i].dbias = db.data_ptr(); probs[i].N = n; probs[i].K = k
    nb = L.iq_wgrad_grouped_ws_bytes(probs, len(shapes), M, 0); ws = torch.empty(nb, dtype=torch.uint8, device=d)
    screen(f"wgrad_big M={M} {shapes}", lambda: L.iq_gemm_bf16_wgrad_grouped(probs, len(shapes), M, ws.data_ptr(), nb, 0, 0, None, 0, st()), outs, perturb)
    del keep, ws
M, D, K = 50432, 192, 768
A = rn(M, K).bfloat16(); W = (rn(D, K) / math.sqrt(K)).bfloat16(); R = rn(M, D).bfloat16(); bias = rn(D); gm = torch.rand(D, device=d) + 0.5; bt = rn(D)
Z = torch.empty(M, D, device=d, dtype=torch.bfloat16); X = torch.empty_like(Z); mean = torch.empty(M, device=d); rstd = torch.empty(M, device=d)
dr = N.Dropout(); dr.p = 0.1; dr.seed = 1; dr.site = 2; dr.step = 3
screen("gemm_ln band 50432x192x768", lambda: L.iq_gemm_bf16_ln(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), R.data_ptr(), D, C.byref(dr), gm.data_ptr(),
                                                               bt.data_ptr(), 1e-12, Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D, K, st()),
       [Z, X, mean, rstd], perturb)
print("race screen:", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
