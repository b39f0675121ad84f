"""In-tree builds of the HIP library (product) and of the host-side SIMT emulator build (tests only)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / 'csrc'
LIB = PKG / 'libry355.so'
EMU_DIR = ROOT / 'tests' / 'emu'
EMU_LIB = EMU_DIR / 'libry355_emu.so'
UNITS = [CSRC / 'ry_net.cpp', CSRC / 'ry_vc.cpp', CSRC / 'ry_comm.cpp']          # translation units of libry355.so
SOURCES = UNITS + [CSRC / 'ry_kernels.h', CSRC / 'ry_vc_kernels.h', CSRC / 'ry_host.h', CSRC / 'ry_dev.h', ROOT / 'include' / 'ry355.h']


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).exists() and Path(d).stat().st_mtime > t for d in deps)


def _hipcc() -> str:
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and Path(c).exists():
            return c
    raise RuntimeError('hipcc not found: cannot build libry355.so')


def build_product(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> realtime_yukarin_amd/libry355.so (cross-compiles without a GPU)."""
    if force or _stale(LIB, SOURCES):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value',
               '-x', 'hip'] + [str(u) for u in UNITS] + ['-o', str(LIB)]
        subprocess.run(cmd, check=True, cwd=str(ROOT))
    return LIB


def build_emu(force: bool = False) -> Path:
    """Same sources as plain C++ on the fiber SIMT emulator (tests/emu) -- test infrastructure only."""
    deps = SOURCES + [EMU_DIR / 'ry_emu.h', EMU_DIR / 'ry_emu.cpp']
    if force or _stale(EMU_LIB, deps):
        cxx = '/opt/rocm/lib/llvm/bin/clang++'
        if not Path(cxx).exists():
            cxx = shutil.which('clang++') or shutil.which('g++')
        cmd = [cxx, '-x', 'c++', '-DRY_HOST_EMU', '-O2', '-std=c++17', '-shared', '-fPIC', '-pthread', '-Wno-psabi',
               '-I' + str(EMU_DIR), '-I' + str(CSRC)] + [str(u) for u in UNITS] + [str(EMU_DIR / 'ry_emu.cpp'), '-o', str(EMU_LIB)]
        subprocess.run(cmd, check=True, cwd=str(ROOT))
    return EMU_LIB


if __name__ == '__main__':
    print(build_product(force=bool(os.environ.get('FORCE'))))
