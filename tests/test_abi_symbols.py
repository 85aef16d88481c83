"""CPU test: the C-ABI library builds for gfx950, loads, and exports every symbol include/atomsmm_hip.h
declares (no compute calls -- there is no GPU here)."""
import os
import re

from atomsmm_amd import backend as B
from atomsmm_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'atomsmm_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(amm_[a-z_0-9]+)\s*\(', text)))


def test_library_builds_and_exports_header_symbols():
    build.build_hip()
    lib = B.lib()
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert missing == []
    assert sorted(B.EXPORTS) == names
    assert lib.amm_abi_version() == 1


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(B.PairDesc) == 4 * 4 + 9 * 8
    assert C.sizeof(B.Op) == 4 * 4 + 8
    assert C.sizeof(B.PairStats) == 4 * 8 + 4 * 4 + 8 + 2 * 4 + 3 * 8 + 8 + 2 * 4 + 2 * 4 + 8 + 2 * 4 + 2 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import pytest
    monkeypatch.setattr(B, '_LIB', None)
    monkeypatch.setattr(B, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(B.HipError, match='no CPU fallback'):
        B.lib()
