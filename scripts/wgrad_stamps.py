"""Diagnostic: where a wgrad workgroup spends its life (scripts/dbg/libwgrad_stamps.so, built with -DIQ_WGRAD_STAMPS).
usage: python scripts/wgrad_stamps.py N K   (env IQ_WGRAD_MC=64|128)"""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "dbg", "libwgrad_stamps.so"))
d = torch.device("cuda:0")
M = 50432
N_, K = int(sys.argv[1]), int(sys.argv[2])
dY = torch.randn(M, N_, device=d).bfloat16(); X = torch.randn(M, K, device=d).bfloat16()
dW = torch.empty(N_, K, device=d); db = torch.empty(N_, device=d)
L.iq_wgrad_ws_bytes.restype = C.c_size_t
L.iq_wgrad_ws_bytes.argtypes = [C.c_int] * 3
nb = L.iq_wgrad_ws_bytes(M, N_, K); ws = torch.empty(nb, dtype=torch.uint8, device=d)
st = torch.zeros(4096 * 6, dtype=torch.int64, device=d)
L.iq_debug_set_wgrad_stamps(C.c_void_p(st.data_ptr()))
L.iq_gemm_bf16_wgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
s = torch.cuda.current_stream().cuda_stream
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N_, K, ws.data_ptr(), nb, 0, s)
a.record()
for _ in range(10):
    L.iq_gemm_bf16_wgrad(dY.data_ptr(), N_, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N_, K, ws.data_ptr(), nb, 0, s)
b.record(); torch.cuda.synchronize()
print(f"N={N_} K={K}: {a.elapsed_time(b) * 100:.1f} us per call (instrumented)")
t = st.cpu().numpy().reshape(-1, 6).astype(np.float64)
t = t[t.sum(1) > 0]
names = (["prologue (first stage)", "frag reads", "MFMA issue", "wait global data", "LDS store + issue loads", "wait for sibling waves"]
         if os.environ.get("IQ_WGRAD_KERNEL", "pw")[0] == "p" else
         ["prologue (first chunk)", "issue next loads", "frag reads + MFMA", "barrier 1", "wait global data", "LDS store + barrier 2"])
tot = t.sum(1).mean()
print(f"{len(t)} workgroups; mean life {tot:.0f} ticks (100 MHz => {tot / 100:.1f} us)")
for i, n in enumerate(names):
    print(f"  {n:26s} mean {t[:, i].mean():8.1f} ticks  {100 * t[:, i].mean() / tot:5.1f}%")
