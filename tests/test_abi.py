"""CPU: the C-ABI library builds, loads, and exports every symbol include/bisinger_hip.h declares."""
import os
import re
import subprocess

from bisinger_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'bisinger_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(bsg_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from bisinger_amd.build import build
    path = build(verbose=False)
    lib = _lib.load()
    assert lib.bsg_abi_version() == _lib.ABI_VERSION
    syms = header_symbols()
    assert len(syms) >= 10
    out = subprocess.check_output(['nm', '-D', '--defined-only', path], text=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for s in syms:
        assert s in exported, f'{s} declared in the header but not exported'
        assert hasattr(lib, s)
    # and the ctypes binding covers the header
    assert sorted(_lib.declared_symbols()) == syms


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    import pytest
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.BsgError):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'bisinger_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith('.py'):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), fn


def test_create_refuses_channel_counts_other_than_256_with_a_message():
    """hparams['residual_channels'] / ['hidden_size'] other than 256 (no shipped config has one): BSG_EINVAL and a message naming the value at
    create, before any device call — checked here without a GPU ("Differences from the reference's surface")."""
    import ctypes
    from ctypes import POINTER, byref, c_void_p, cast
    lib = _lib.load()
    dummy = (c_void_p * 200)(*([1] * 200))
    h = c_void_p()
    cfg = _lib.DiffnetCfg(80, 384, 384, 20, 4, 1000)
    rc = lib.bsg_diffnet_create(byref(h), byref(cfg), cast(dummy, POINTER(c_void_p)), 170, c_void_p(1), None)
    assert rc != 0 and h.value is None
    msg = lib.bsg_last_error().decode()
    assert '384' in msg and '256' in msg, msg
    fcfg = _lib.Fs2Cfg(192, 65, 4, 4, 2, 9, 9, 80, 2, 3, 22, 8, 2002, 5000)
    rc = lib.bsg_fs2midi_create(byref(h), byref(fcfg), cast(dummy, POINTER(c_void_p)), 143, c_void_p(1), c_void_p(1), None)
    assert rc != 0 and h.value is None
    msg = lib.bsg_last_error().decode()
    assert '192' in msg and '256' in msg, msg
