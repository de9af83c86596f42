import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["gemm_bench.py"]
src = open("scripts/gemm_bench.py").read()
src = src[:src.index('print(f"M = {M}")')]
exec(src)
nt(192, 768)
nt(192, 768, bias=True)
nt(192, 768, bias=True, drop=0.1)
nt(192, 768, bias=True, drop=0.1, res=True)
nt(192, 768, res=True)
nt(192, 576)
nt(192, 576, res=True)
