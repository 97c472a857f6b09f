"""Runs bench.py against another build of the library (same-box A/B of the whole step in one gpurun call):
    python tools/ab_step.py path/to/libp4p_hip.so [bench.py arguments]"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import p4p_amd
from p4p_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
