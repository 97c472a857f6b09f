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


_SCALARS = {'int': ctypes.c_int, 'long long': ctypes.c_longlong, 'float': ctypes.c_float, 'double': ctypes.c_double,
            'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32, 'uint32_t': ctypes.c_uint32}


def _ctype_of(decl):
    """ctypes type of one parameter / return declaration of the header ('const float* x', 'long long n', 'void** out', ...).
    Every pointer is passed as an address (c_void_p: an int, None, a c_void_p or a ctypes array / byref)."""
    d = re.sub(r'\bconst\b', ' ', decl).strip()
    if '*' in d:
        return ctypes.c_char_p if re.fullmatch(r'char\s*\*', d) else ctypes.c_void_p
    d = re.sub(r'\s+', ' ', d)
    for name in sorted(_SCALARS, key=len, reverse=True):          # 'long long n' before 'int n'
        if d == name or d.startswith(name + ' '):
            return _SCALARS[name]
    raise P4PError(f'{HEADER_PATH}: cannot map the C declaration {decl!r} to ctypes')


def declared_signatures():
    """{entry point: (restype, [argtypes])} parsed from the prototypes of include/p4p_hip.h.  The binding sets these on
    every function it calls, so a Python int reaches a `long long` parameter as 64 bits and a float can never land in an
    integer slot silently (ctypes' default is c_int for every int and a refusal for floats without argtypes)."""
    text = re.sub(r'/\*.*?\*/', '', _header_text(), flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    sigs = {}
    for ret, name, args in re.findall(r'\b((?:const\s+)?[a-z_0-9]+(?:\s+long)?\s*\**)\s*(p4p_[a-z0-9_]+)\s*\(([^()]*)\)\s*;', text):
        args = args.strip()
        params = [] if args in ('', 'void') else [a.strip() for a in args.split(',')]
        # a parameter is '<type> <name>'; the type of the RETURN has no name behind it
        sigs[name] = (_ctype_of(ret.strip()) if '*' in ret else _ctype_of(ret.strip() + ' _'),
                      [_ctype_of(a) for a in params])
    return sigs


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
        missing = []
        for name, (res, args) in declared_signatures().items():
            fn = getattr(cand, name, None)
            if fn is None:
                missing.append(name)
                continue
            fn.restype, fn.argtypes = res, args
        if missing:
            raise P4PError(f'{LIB_PATH} does not export {missing} declared in include/p4p_hip.h: rebuild it')
        _lib = cand
    return _lib


def _fn(name):
    fn = getattr(lib(), name)
    if fn.argtypes is None:           # not declared in the header (an experiment hook of tools/): no silent c_int guessing
        raise P4PError(f'{name} is exported by the library but not declared in include/p4p_hip.h: declare it there '
                       '(the binding takes every argument type from the header)')
    return fn


def call(name, *args):
    """Status-returning entry point: raises P4PError with the library's message on a non-zero return.  Arguments are converted
    by the argtypes taken from the header: ints to the declared width, floats to float / double, None / int / c_void_p to pointers."""
    try:
        rc = _fn(name)(*args)
    except ctypes.ArgumentError as e:
        raise P4PError(f'{name}: argument does not fit the prototype in include/p4p_hip.h ({e})') from e
    if rc != 0:
        raise P4PError(f'{name}: {lib().p4p_last_error().decode()}')


def call_int(name, *args):
    """Entry point whose int return IS the result (sizes, previous switch values)."""
    try:
        return int(_fn(name)(*args))
    except ctypes.ArgumentError as e:
        raise P4PError(f'{name}: argument does not fit the prototype in include/p4p_hip.h ({e})') from e


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def i64(v):
    return ctypes.c_longlong(int(v))
