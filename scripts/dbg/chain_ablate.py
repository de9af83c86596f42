import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import vit_vs_raw_iq_amd._native as N
N.LIB_PATH = os.path.join(root, "scripts", "dbg", sys.argv[1])
sys.argv = ["x"]
exec(open(os.path.join(root, "scripts", "chain_bench.py")).read())
