"""Every kernel call of one encoder layer's training step, timed alone at a named geometry (HIP events, 20 launches each):
     python scripts/layer_kernels.py D H F S B [drop]
e.g. cfg C: 128 8 1024 65 256 0.2   cfg B: 192 3 768 197 256 0.1   cfg D: 768 12 3072 197 512 0.1
The sum is what a layer costs with every launch back to back on a warm chip; the step adds launch gaps."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
D, H, F, S, B = (int(v) for v in sys.argv[1:6])
drop = float(sys.argv[6]) if len(sys.argv) > 6 else 0.1
M, dh = B * S, D // H
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
bf = lambda *s: torch.randn(*s, device=d).bfloat16()
def dr(site):
    x = N.Dropout(); x.p = drop; x.seed = 1; x.site = site; x.step = 3
    return x
rows = []
def rec(name, us, byt, fl):
    rows.append((name, us, byt, fl))
    print(f"{name:34s} {us:8.1f} us  {byt / us / 1e3:8.1f} GB/s  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
def nt(name, N_, K, bias=False, relu=False, dropsite=0, res=False, gate=False):
    A = bf(M, K); W = (torch.randn(N_, K, device=d) / math.sqrt(K)).bfloat16(); Cc = torch.empty(M, N_, device=d, dtype=torch.bfloat16)
    e = N.Epilogue(); keep = []
    if bias: t = torch.randn(N_, device=d); keep.append(t); e.bias = t.data_ptr()
    e.relu = 1 if relu else 0
    if dropsite and drop > 0: e.drop = dr(dropsite)
    if res: t = bf(M, N_); keep.append(t); e.residual = t.data_ptr(); e.ldr = N_
    if gate: t = bf(M, N_); keep.append(t); e.gate = t.data_ptr(); e.ldg = N_; e.gate_scale = 1.1
    us = timeit(lambda: L.iq_gemm_bf16_nt(A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), st()))
    rec(name, us, 2 * (M * K + N_ * K + M * N_) + (2 * M * N_ if res else 0) + (2 * M * N_ if gate else 0), 2 * M * N_ * K)
def gemm_ln(name, K, site):
    A = bf(M, K); W = (torch.randn(D, K, device=d) / math.sqrt(K)).bfloat16(); R = bf(M, D)
    bias = torch.randn(D, device=d); gm = torch.rand(D, device=d) + 0.5; bt = torch.randn(D, device=d)
    Z = torch.empty(M, D, device=d, dtype=torch.bfloat16); X = torch.empty_like(Z); mean = torch.empty(M, device=d); rstd = torch.empty(M, device=d)
    x = dr(site)
    if not L.iq_gemm_ln_supported(D, K):
        return nt(name + " (unfused)", D, K, bias=True, dropsite=site, res=True)
    us = timeit(lambda: L.iq_gemm_bf16_ln(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), R.data_ptr(), D, C.byref(x) if drop > 0 else None,
                                          gm.data_ptr(), bt.data_ptr(), 1e-12, Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D, K, st()))
    rec(name, us, 2 * (M * K + D * K + 3 * M * D) + 8 * M, 2 * M * D * K)
def gemm_lnbwd(name, K, site):
    if not L.iq_gemm_lnbwd_supported(D, K):
        return nt(name + " (unfused)", D, K, res=True)
    A = bf(M, K); W = (torch.randn(D, K, device=d) / math.sqrt(K)).bfloat16(); R = bf(M, D); z = bf(M, D)
    mean = z.float().mean(-1).contiguous(); rstd = (1 / torch.sqrt(z.float().var(-1, unbiased=False) + 1e-12)).contiguous()
    gm = torch.rand(D, device=d) + 0.5; dz = torch.empty_like(z); dy = torch.empty_like(z)
    part = torch.empty(L.iq_gemm_lnbwd_partial_rows(M), 2 * D, device=d)
    x = dr(site)
    us = timeit(lambda: L.iq_gemm_bf16_lnbwd(A.data_ptr(), K, W.data_ptr(), K, R.data_ptr(), D, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             gm.data_ptr(), C.byref(x) if drop > 0 else None, dz.data_ptr(), dy.data_ptr(), part.data_ptr(), M, D, K, st()))
    rec(name, us, 2 * (M * K + D * K + M * D * (3 + (1 if drop > 0 else 0))) + 8 * M, 2 * M * D * K)
def attn():
    qkv = bf(M, 3 * D); out = torch.empty(M, D, device=d, dtype=torch.bfloat16); lse = torch.empty(B, H, S, device=d)
    dout = bf(M, D); dqkv = torch.empty_like(qkv)
    tf = timeit(lambda: L.iq_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, dh, st()))
    rec("attention fwd", tf, 2 * M * 4 * D, 4 * B * H * S * S * dh)
    tb = timeit(lambda: L.iq_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, S, H, dh, st()))
    return tb
def wgrad():
    probs = (N.WgradProblem * 4)()
    keep = []
    for i, (n_, k_) in enumerate(((D, F), (F, D), (D, D), (3 * D, D))):
        dY = bf(M, n_); X = bf(M, k_); dW = torch.empty(n_, k_, device=d); db = torch.empty(n_, device=d)
        keep += [dY, X, dW, db]
        probs[i].dY = dY.data_ptr(); probs[i].ldy = n_; probs[i].X = X.data_ptr(); probs[i].ldx = k_
        probs[i].dW = dW.data_ptr(); probs[i].dbias = db.data_ptr(); probs[i].N = n_; probs[i].K = k_
    nb = L.iq_wgrad_grouped_ws_bytes(probs, 4, M, 0); ws = torch.empty(nb, dtype=torch.uint8, device=d)
    us = timeit(lambda: L.iq_gemm_bf16_wgrad_grouped(probs, 4, M, ws.data_ptr(), nb, 0, 0, None, 0, st()))
    nk = D * F * 2 + D * D * 4
    rec("weight gradients (grouped + reduce)", us, 2 * M * (2 * D + 2 * F + 2 * D + 4 * D) + 4 * nk, 2 * M * nk)
def ffn_chain():
    if not L.iq_ffn_chain_supported(S, D, F):
        return
    X1 = bf(M, D); W1 = (torch.randn(F, D, device=d) / math.sqrt(D)).bfloat16(); W2 = (torch.randn(D, F, device=d) / math.sqrt(F)).bfloat16()
    b1 = torch.randn(F, device=d); b2 = torch.randn(D, device=d); gm = torch.rand(D, device=d) + 0.5; bt = torch.randn(D, device=d)
    Hh = torch.empty(M, F, device=d, dtype=torch.bfloat16); Z = torch.empty(M, D, device=d, dtype=torch.bfloat16); X = torch.empty_like(Z)
    mean = torch.empty(M, device=d); rstd = torch.empty(M, device=d)
    d1, d2 = dr(2), dr(3)
    us = timeit(lambda: L.iq_ffn_chain_fwd(X1.data_ptr(), W1.data_ptr(), b1.data_ptr(), C.byref(d1) if drop > 0 else None, Hh.data_ptr(),
                                           W2.data_ptr(), b2.data_ptr(), C.byref(d2) if drop > 0 else None, gm.data_ptr(), bt.data_ptr(), 1e-12,
                                           Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, B, S, D, F, st()))
    print(f"{'[ffn chain: ffn1 + ffn2 + norm2]':34s} {us:8.1f} us  {(2 * (3 * M * D + M * F + 2 * D * F)) / us / 1e3:8.1f} GB/s  {4 * M * D * F / us / 1e6:8.1f} TFLOP/s", flush=True)
    A = bf(M, D); R = bf(M, D); Wo = (torch.randn(D, D, device=d) / math.sqrt(D)).bfloat16(); bo = torch.randn(D, device=d)
    Z1 = torch.empty_like(Z); m1 = torch.empty(M, device=d); r1 = torch.empty(M, device=d); d0 = dr(1)
    gate = torch.empty(L.iq_ffn_chain_gate_bytes(M, F) // 4, dtype=torch.int32, device=d)
    dp = lambda x: C.byref(x) if drop > 0 else None
    us = timeit(lambda: L.iq_attn_out_ffn_chain_fwd(A.data_ptr(), Wo.data_ptr(), bo.data_ptr(), dp(d0), R.data_ptr(), gm.data_ptr(), bt.data_ptr(),
                                                    Z1.data_ptr(), X1.data_ptr(), m1.data_ptr(), r1.data_ptr(), W1.data_ptr(), b1.data_ptr(), dp(d1),
                                                    Hh.data_ptr(), W2.data_ptr(), b2.data_ptr(), dp(d2), gm.data_ptr(), bt.data_ptr(), 1e-12,
                                                    Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gate.data_ptr(), None, None, None,
                                                    B, S, D, F, st()))
    print(f"{'[out-proj + norm1 + ffn chain]':34s} {us:8.1f} us  (gate bits written)", flush=True)
    Wq = (torch.randn(3 * D, D, device=d) / math.sqrt(D)).bfloat16(); bq = torch.randn(3 * D, device=d); Yq = torch.empty(M, 3 * D, device=d, dtype=torch.bfloat16)
    us = timeit(lambda: L.iq_attn_out_ffn_chain_fwd(A.data_ptr(), Wo.data_ptr(), bo.data_ptr(), dp(d0), R.data_ptr(), gm.data_ptr(), bt.data_ptr(),
                                                    Z1.data_ptr(), X1.data_ptr(), m1.data_ptr(), r1.data_ptr(), W1.data_ptr(), b1.data_ptr(), dp(d1),
                                                    Hh.data_ptr(), W2.data_ptr(), b2.data_ptr(), dp(d2), gm.data_ptr(), bt.data_ptr(), 1e-12,
                                                    Z.data_ptr(), X.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gate.data_ptr(), Wq.data_ptr(),
                                                    bq.data_ptr(), Yq.data_ptr(), B, S, D, F, st()))
    print(f"{'[... + next q,k,v]':34s} {us:8.1f} us", flush=True)
print(f"D={D} H={H} F={F} S={S} B={B} M={M} drop={drop}")
nt("qkv projection", 3 * D, D, bias=True)
tb = attn()
gemm_ln("out-proj + drop + res + norm1", D, 1)
nt("ffn1 + relu + drop", F, D, bias=True, relu=True, dropsite=2)
gemm_ln("ffn2 + drop + res + norm2", F, 3)
ffn_chain()
fwd = sum(r[1] for r in rows)
nt("ffn2 dgrad (gate)", F, D, gate=True)
gemm_lnbwd("ffn1 dgrad + norm1 bwd", F, 1)
nt("out-proj dgrad", D, D)
rec("attention bwd", tb, 2 * M * 8 * D, 14 * B * H * S * S * dh)
wgrad()
gemm_lnbwd("qkv dgrad + norm2 bwd", 3 * D, 3)
tot = sum(r[1] for r in rows)
print(f"layer forward {fwd:.1f} us, backward {tot - fwd:.1f} us, total {tot:.1f} us;  "
      f"{sum(r[2] for r in rows) / tot / 1e3:.0f} GB/s, {sum(r[3] for r in rows) / tot / 1e6:.0f} TFLOP/s over the layer")
