"""The C-ABI library loads and exports every symbol include/flowmol_hip.h declares (no compute: no GPU here)."""
import ctypes
import re
import shutil
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    txt = (ROOT / 'include' / 'flowmol_hip.h').read_text()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(fm_[a-z_0-9]+)\s*\(', txt)))


def test_header_declares_the_boundary():
    fns = header_functions()
    for must in ('fm_create', 'fm_destroy', 'fm_batch_bind', 'fm_forward', 'fm_ctmc_step', 'fm_integrate', 'fm_last_error'):
        assert must in fns


def test_library_builds_and_exports_every_declared_symbol():
    from flowmol_amd import _lib, build
    if not (shutil.which('hipcc') or Path('/opt/rocm/bin/hipcc').exists() or build.OUT.exists()):
        pytest.skip('no hipcc and no prebuilt library')
    lib_path = build.build(verbose=False)
    lib = ctypes.CDLL(str(lib_path))
    for fn in header_functions():
        assert hasattr(lib, fn), f'{fn} declared in include/flowmol_hip.h but not exported'
    typed = _lib.load(lib_path)
    assert typed.fm_abi_version() == _lib.FM_ABI_VERSION
    assert set(_lib.EXPORTED_SYMBOLS) == set(header_functions())


def test_missing_library_fails_loudly(tmp_path):
    from flowmol_amd import _lib
    with pytest.raises(_lib.FlowMolHipError, match='no CPU fallback'):
        _lib.load(tmp_path / 'libflowmol_hip.so')


def test_product_never_imports_the_oracle():
    for py in (ROOT / 'flowmol_amd').rglob('*.py'):
        src = py.read_text()
        assert 'import oracle' not in src and 'from oracle' not in src and 'cpu_ref' not in src, py
