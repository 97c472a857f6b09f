"""Runs bench.py with module attributes of the package overridden (same-box A/B of a host-side change in one gpurun call):
    python tools/ab_flag.py functional._FUSE_QKV_BIAS_MAXT=0 [bench.py arguments]"""
import importlib, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import p4p_amd
args = sys.argv[1:]
while args and '=' in args[0] and not args[0].startswith('-'):
    path, val = args.pop(0).split('=', 1)
    mod, attr = path.rsplit('.', 1)
    m = importlib.import_module('p4p_amd.' + mod)
    old = getattr(m, attr)
    setattr(m, attr, type(old)(int(val)) if isinstance(old, (bool, int)) else type(old)(val))
    print(f'[ab_flag] p4p_amd.{mod}.{attr}: {old!r} -> {getattr(m, attr)!r}', file=sys.stderr)
sys.argv = [os.path.join(root, 'bench.py')] + args
runpy.run_path(sys.argv[0], run_name='__main__')
