"""Per-launch floor inside a hipGraph: a chain of dependent tiny kernels (LayerNorm on 64 rows, a 1-element add) replayed
from a graph vs eager.  Tells how much of the ~6 us intercept of the M-sweeps is launch / dependency overhead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_vs_raw_iq_amd._native as N
L = N.lib(); d = torch.device("cuda:0")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
def chain(kind, n, M):
    D = 192
    z = torch.randn(M, D, device=d).bfloat16(); x = torch.empty_like(z)
    g = torch.ones(D, device=d); b = torch.zeros(D, device=d); mean = torch.empty(M, device=d); rstd = torch.empty(M, device=d)
    t = torch.zeros(1, device=d)
    s = torch.cuda.Stream()
    def body(stream):
        for i in range(n):
            if kind == "ln":
                src, dst = (z, x) if i % 2 == 0 else (x, z)
                L.iq_ln_fwd(src.data_ptr(), g.data_ptr(), b.data_ptr(), dst.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D, 1e-12, stream)
            else:
                L.iq_counter_add(t.data_ptr(), 1, None, 0.0, stream) if False else t.add_(1)
    with torch.cuda.stream(s):
        body(s.cuda_stream); torch.cuda.synchronize()
        eager = timed(lambda: body(s.cuda_stream)) / n
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            body(s.cuda_stream)
        graph = timed(gr.replay) / n
    return eager, graph
for kind, M in (("add", 1), ("ln", 64), ("ln", 4096), ("ln", 50432)):
    e, g_ = chain(kind, 100, M)
    print(f"{kind:4s} M={M:6d}: eager {e:6.2f} us/launch   graph replay {g_:6.2f} us/launch")
