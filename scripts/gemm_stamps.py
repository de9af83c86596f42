"""Diagnostic: per-workgroup phase timestamps of the async NT GEMM (scripts/dbg/libgemm_stamps.so, built with -DIQ_GEMM_STAMPS)."""
import ctypes as C, os, sys, math
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import vit_vs_raw_iq_amd._native as N
L = C.CDLL(os.path.join(here, "dbg", "libgemm_stamps.so"))
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
bn = 128 if (N_ % 128 == 0 or N_ > 512) else 64
grid = ((M + 127) // 128) * ((N_ + bn - 1) // bn)
st = torch.zeros(grid * 6, dtype=torch.int64, device=d)
L.iq_debug_set_stamps(C.c_void_p(st.data_ptr()))
L.iq_gemm_bf16_nt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.iq_gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N_, M, N_, K, C.byref(e), s)
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(grid, 6).astype(np.float64)
t0 = t[:, 0].min()
clk = 100e6   # s_memtime ticks at 100 MHz? print raw + ratios
span = t[:, 5].max() - t0
print(f"grid {grid} WGs; kernel span {span:.0f} ticks")
names = ["start->first stage landed", "K loop", "barrier", "epilogue issue", "store drain"]
for i, n in enumerate(names):
    dlt = t[:, i + 1] - t[:, i]
    print(f"  {n:28s} mean {dlt.mean():9.1f}  p50 {np.median(dlt):9.1f}  p90 {np.percentile(dlt, 90):9.1f}  ({100 * dlt.mean() / (t[:,5]-t[:,0]).mean():5.1f}% of WG life)")
life = t[:, 5] - t[:, 0]
print(f"  WG life mean {life.mean():.0f} ticks; start spread: p10 {np.percentile(t[:,0]-t0,10):.0f} p50 {np.percentile(t[:,0]-t0,50):.0f} p90 {np.percentile(t[:,0]-t0,90):.0f}")
print(f"  concurrency = sum(life)/span = {life.sum()/span:.1f} WGs in flight on average (256 CUs)")
