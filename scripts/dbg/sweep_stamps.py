"""Diagnostic: where a wave of the A-stationary sweep GEMM (scripts/dbg/variants/gemm_sweep.hip, not in the library)
spends its life.  Needs scripts/dbg/libgemm_stamps.so = hipcc -DIQ_GEMM_STAMPS -shared gemm_nt.hip gemm_sweep.hip prof.hip
with the variant copied next to gemm_nt.hip and its hook (gemm_sweep_try) restored in iq_gemm_bf16_nt.
usage: python scripts/dbg/sweep_stamps.py N K [bias] [relu] [drop] [gate]"""
import ctypes as C, os, sys, math
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import vit_vs_raw_iq_amd._native as N
L = C.CDLL(os.path.join(here, "libgemm_stamps.so"))
d = torch.device("cuda:0")
M = 50432
N_, K = int(sys.argv[1]), int(sys.argv[2])
flags = sys.argv[3:]
A = torch.randn(M, K, device=d).bfloat16(); B = (torch.randn(N_, K, device=d) / math.sqrt(K)).bfloat16()
Cc = torch.empty(M, N_, device=d, dtype=torch.bfloat16)
e = N.Epilogue(); keep = []
if "bias" in flags: t = torch.randn(N_, device=d); keep.append(t); e.bias = t.data_ptr()
if "relu" in flags: e.relu = 1
if "drop" in flags: e.drop.p = 0.1; e.drop.seed = 1
if "gate" in flags: t = torch.randn(M, N_, device=d).bfloat16(); keep.append(t); e.gate = t.data_ptr(); e.ldg = N_; e.gate_scale = 1.1
nw = 512 * 4
st = torch.zeros(nw * 6, dtype=torch.int64, device=d)
L.iq_debug_set_stamps(C.c_void_p(st.data_ptr()))
L.iq_gemm_bf16_nt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), s)
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(nw, 6).astype(np.float64)
t0 = t[:, 0].min()
life = t[:, 5] - t[:, 0]
print(f"{N_}x{K} {flags}: kernel span {t[:,5].max() - t0:.0f} ticks (100 MHz); wave life mean {life.mean():.0f} p90 {np.percentile(life,90):.0f}; start spread p90 {np.percentile(t[:,0]-t0,90):.0f}")
for name, v in (("A block landed", t[:, 1] - t[:, 0]), ("stage waits (sum)", t[:, 2]), ("stage compute (sum)", t[:, 3]), ("epilogues (sum)", t[:, 4]),
                ("final store drain+rest", life - (t[:, 1] - t[:, 0]) - t[:, 2] - t[:, 3] - t[:, 4])):
    print(f"  {name:24s} mean {v.mean():8.1f}  p50 {np.median(v):8.1f}  p90 {np.percentile(v, 90):8.1f}   {100 * v.mean() / life.mean():5.1f} % of wave life")
