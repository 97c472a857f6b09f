"""ctypes binding of libp4p_hip.so (the C ABI declared in include/p4p_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C <package>/csrc``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libp4p_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'p4p_hip.h')

_lib = None


class P4PError(RuntimeError):
    pass


def _header_text():
    try:
        return open(HEADER_PATH).read()
    except OSError as e:
        raise P4PError(f'{HEADER_PATH} cannot be read ({e}): the package is deployed without include/p4p_hip.h, whose '
                       'P4P_ABI_VERSION the loader compares with the library') from e


def declared_symbols():
    """Entry points named in include/p4p_hip.h (used by the CPU test that checks the exports)."""
    text = _header_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(p4p_[a-z0-9_]+)\s*\(', text)))


def header_abi_version():
    """P4P_ABI_VERSION of include/p4p_hip.h: the ABI this Python package was written against."""
    m = re.search(r'#define\s+P4P_ABI_VERSION\s+(\d+)', _header_text())
    if m is None:
        raise P4PError(f'{HEADER_PATH}: P4P_ABI_VERSION not found')
    return int(m.group(1))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P4PError(f'{LIB_PATH} not found: the HIP library has not been built '
                           '(run __graft_entry__.build()); there is no fallback path')
        cand = ctypes.CDLL(LIB_PATH)
        cand.p4p_last_error.restype = ctypes.c_char_p
        cand.p4p_abi_version.restype = ctypes.c_int
        have, want = int(cand.p4p_abi_version()), header_abi_version()
        if have != want:
            raise P4PError(f'{LIB_PATH} implements ABI version {have}, include/p4p_hip.h declares {want}: the library is '
                           'stale (argument lists differ) -- rebuild it with __graft_entry__.build()')
        _lib = cand
    return _lib


def _arg(v):
    if v is None:
        return ctypes.c_void_p(0)
    if isinstance(v, bool):
        return ctypes.c_int(int(v))
    if isinstance(v, int):
        return ctypes.c_int(v)
    if isinstance(v, float):
        return ctypes.c_float(v)
    return v


def call(name, *args):
    fn = getattr(lib(), name)
    rc = fn(*[_arg(a) for a in args])
    if rc != 0:
        raise P4PError(f'{name}: {lib().p4p_last_error().decode()}')


def call_int(name, *args):
    fn = getattr(lib(), name)
    fn.restype = ctypes.c_int
    return int(fn(*[_arg(a) for a in args]))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def i64(v):
    return ctypes.c_longlong(int(v))
