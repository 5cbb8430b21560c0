import sys, os, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import itsxpress_amd._lib as l
l.LIB_PATH = os.path.join(ROOT, "build", sys.argv[1], "libitsx_hip.so")
sys.argv = ["cluster_curve.py", "--sizes", sys.argv[2]]
runpy.run_path(os.path.join(ROOT, "scripts", "cluster_curve.py"), run_name="__main__")
