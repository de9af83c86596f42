"""Run scripts/gemm_bench.py's NT cases against an ablated build of the library (timing only; results are wrong).
usage: IQ_GEMM_WP=1 python scripts/dbg/ablate.py <libname>   (libs: hipcc -DIQ_WP_NO_MFMA | -DIQ_WP_NO_DMA | -DIQ_EPI_NO_STORE)"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import vit_vs_raw_iq_amd._native as N
N.LIB_PATH = os.path.join(root, "scripts", "dbg", sys.argv[1])
sys.argv = ["gemm_bench.py"]
exec(open(os.path.join(root, "scripts", "gemm_bench.py")).read())
