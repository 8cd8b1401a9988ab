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
