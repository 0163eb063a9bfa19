import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return ROOT / 'tests' / 'golden'


EMU = ROOT / 'tests' / 'emu' / 'libflowmol_emu.so'


@pytest.fixture(scope='session')
def emu_lib_path():
    """Build (if stale) the host emulation of the HIP kernels: the SAME kernel and host sources compiled with the host
    clang++ and run lane-by-lane on the CPU (tests/emu).  Test tooling only; never loaded by the product."""
    import shutil
    import subprocess
    cxx = '/opt/rocm/lib/llvm/bin/clang++'
    if not Path(cxx).exists() and not shutil.which('clang++'):
        pytest.skip('no clang++ to build the host emulation')
    here = ROOT / 'tests'
    srcs = [here / 'emu' / 'emu_rt.cpp', here / 'emu' / 'hip' / 'hip_runtime.h', ROOT / 'include' / 'flowmol_hip.h'] + \
        sorted((ROOT / 'flowmol_amd' / 'csrc').glob('*'))
    if not EMU.exists() or any(s_.stat().st_mtime > EMU.stat().st_mtime for s_ in srcs):
        subprocess.run([str(here / 'emu' / 'build_emu.sh')], check=True, capture_output=True)
    return EMU


@pytest.fixture(scope='session')
def emu_lib(emu_lib_path):
    from flowmol_amd import _lib
    return _lib.load(emu_lib_path)
