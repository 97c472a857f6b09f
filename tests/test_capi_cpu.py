"""CPU: the C-ABI library loads and exports every symbol include/p4p_hip.h declares; argument
validation fails loudly without touching a GPU (no compute calls here)."""
import ctypes
import os

import pytest
import torch

import p4p_amd
from p4p_amd import _lib, ops


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    assert os.path.dirname(_lib.LIB_PATH).startswith(os.path.dirname(os.path.abspath(p4p_amd.__file__)))


def test_every_declared_symbol_is_exported():
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/p4p_hip.h but not exported'
    assert lib.p4p_abi_version() == _lib.header_abi_version() >= 3


def test_a_stale_library_is_refused(monkeypatch):
    """the .so is built apart from the package; a build with another argument list must not be bound."""
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'header_abi_version', lambda: 999)
    with pytest.raises(_lib.P4PError, match='stale'):
        _lib.lib()


def test_argument_checks_fail_loudly_without_launching():
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    rc = lib.p4p_fps_f32(null, null, null, 1, 16, 4, null)
    assert rc == 1 and b'null pointer' in lib.p4p_last_error()
    one = ctypes.c_void_p(64)      # never dereferenced: size checks come first
    rc = lib.p4p_knn_f32(one, one, one, 1, 8, 2, 9, null)
    assert rc == 1 and b'exceeds N' in lib.p4p_last_error()
    rc = lib.p4p_gemm_tn_bf16(0, one, one, one, null, null, null, null, null, 128, 128, 100, 1, null)
    assert rc == 1 and b'multiple of 64' in lib.p4p_last_error()
    rc = lib.p4p_attention_fwd(one, one, null, 1, 16, 3, 100, ctypes.c_float(0.125), null)
    assert rc == 1 and b'head_dim 64' in lib.p4p_last_error()


def test_cpu_tensors_are_rejected_not_silently_computed():
    pts = torch.zeros(2, 64, 3)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.fps(pts, 16)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.knn(pts, pts[:, :4].contiguous(), 4)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.gemm_tn(torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16))


def test_wgrad_split_heuristic_is_sane():
    lib = _lib.lib()
    for M, N, K in [(65792, 3072, 768), (65792, 768, 768), (2056, 576, 192), (64, 128, 128)]:
        s = lib.p4p_wgrad_splits(M, N, K)
        assert 1 <= s <= 64 and s <= (M + 63) // 64


def test_every_declared_symbol_has_its_prototype_bound():
    """the binding takes restype / argtypes of every entry point from include/p4p_hip.h, so a Python int
    reaches a `long long` parameter as 64 bits and a float cannot land in an integer slot."""
    lib = _lib.lib()
    sigs = _lib.declared_signatures()
    assert sorted(sigs) == _lib.declared_symbols()
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        assert fn.restype is res and list(fn.argtypes) == args, name
    assert sigs['p4p_last_error'] == (ctypes.c_char_p, [])
    assert sigs['p4p_sum_slabs_f32'][1] == [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    assert sigs['p4p_clip_coef'][1][2:4] == [ctypes.c_float, ctypes.c_float]
    assert sigs['p4p_pos_mlp0_bwd_rows'] == (ctypes.c_int, [ctypes.c_longlong])
    assert len(sigs['p4p_gemm_tn_bf16'][1]) == 14 and len(sigs['p4p_embed_stage1_bwd'][1]) == 27


def test_arguments_that_do_not_fit_the_prototype_are_refused():
    # a 64-bit count passes through a `long long` parameter unharmed (bare Python int, no i64() wrapper) ...
    assert _lib.call_int('p4p_pos_mlp0_bwd_rows', 5_000_000_000) > 1_000_000
    # ... a float in an int slot and a wrong argument count are errors, not silent reinterpretations
    with pytest.raises(_lib.P4PError, match='prototype'):
        _lib.call_int('p4p_wgrad_splits', 1024.0, 128, 128)
    with pytest.raises(TypeError):
        _lib.call_int('p4p_wgrad_splits', 1024, 128)
    # an export the header does not declare is not callable through the binding
    with pytest.raises((_lib.P4PError, AttributeError)):
        _lib.call('p4p_not_an_entry_point')
