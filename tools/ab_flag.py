"""Runs bench.py with module attributes of the package overridden, or a library switch set (same-box A/B in one gpurun call):
    python tools/ab_flag.py functional._FUSE_QKV_BIAS_MAXT=0 [bench.py arguments]
    python tools/ab_flag.py lib.p4p_set_wgrad_wide=6 [bench.py arguments]"""
import importlib, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import p4p_amd
args = sys.argv[1:]
while args and '=' in args[0] and not args[0].startswith('-'):
    path, val = args.pop(0).split('=', 1)
    mod, attr = path.rsplit('.', 1)
    if mod == 'lib':                                   # a p4p_set_* switch of the C library
        old = p4p_amd._lib.call_int(attr, int(val))
        print(f'[ab_flag] {attr}({val}) (was {old})', file=sys.stderr)
        continue
    m = importlib.import_module('p4p_amd.' + mod)
    old = getattr(m, attr)
    setattr(m, attr, type(old)(int(val)) if isinstance(old, (bool, int)) else type(old)(val))
    print(f'[ab_flag] p4p_amd.{mod}.{attr}: {old!r} -> {getattr(m, attr)!r}', file=sys.stderr)
sys.argv = [os.path.join(root, 'bench.py')] + args
runpy.run_path(sys.argv[0], run_name='__main__')
