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
UNITS = [CSRC / 'ry_exec.cpp', CSRC / 'ry_plan.cpp', CSRC / 'ry_net.cpp', CSRC / 'ry_vc.cpp', CSRC / 'ry_comm.cpp']          # translation units of libry355.so (ry_exec.cpp carries the kernels)
SOURCES = UNITS + [CSRC / 'ry_kernels.h', CSRC / 'ry_vc_kernels.h', CSRC / 'ry_plan.h', CSRC / 'ry_host.h', CSRC / 'ry_dev.h', ROOT / 'include' / 'ry355.h']


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


def _compile_units(cc, flags, obj_dir: Path, suffix: str, extra_sources=()):
    """Every translation unit to its own object file, side by side (ry_exec.cpp carries the kernel instantiations and takes most of
    the time), then the caller links them."""
    from concurrent.futures import ThreadPoolExecutor
    obj_dir.mkdir(parents=True, exist_ok=True)
    jobs = [(u, obj_dir / (Path(u).stem + suffix + '.o')) for u in list(UNITS) + list(extra_sources)]

    def one(job):
        src, obj = job
        tmp = obj.with_suffix('.o.tmp%d' % os.getpid())                    # several builders may run at once (xdist workers): nobody links a
        subprocess.run([cc] + flags + ['-c', str(src), '-o', str(tmp)], check=True, cwd=str(ROOT))   # half-written object of somebody else
        os.replace(str(tmp), str(obj))
        return obj
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        return [str(o) for o in ex.map(one, jobs)]


def build_product(force: bool = False, defs=(), suffix: str = '') -> Path:
    """hipcc --offload-arch=gfx950 -> realtime_yukarin_amd/libry355.so (cross-compiles without a GPU).  `defs` / `suffix`: an experiment build
    with extra -D switches next to the product (`libry355<suffix>.so`, e.g. for an A/B of a compile-time variant on the GPU box)."""
    if defs and not suffix:
        raise ValueError('an experiment build (extra -D switches) needs a suffix: it must never replace the product library')
    lib = LIB if not suffix else LIB.with_name('libry355%s.so' % suffix)
    if force or _stale(lib, SOURCES):
        cc = _hipcc()
        objs = _compile_units(cc, ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value'] + ['-D' + d for d in defs] + ['-x', 'hip'],
                              ROOT / 'gpurun_out' / '_obj', suffix)
        tmp = lib.with_suffix('.so.tmp%d' % os.getpid())                   # link aside, then rename: a process that has the old library mapped keeps it
        subprocess.run([cc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', str(tmp)], check=True, cwd=str(ROOT))
        os.replace(str(tmp), str(lib))
    return lib


def build_emu(force: bool = False, defs=(), suffix: str = '', sanitize: bool = False) -> Path:
    """Same sources as plain C++ on the fiber SIMT emulator (tests/emu) -- test infrastructure only.
    sanitize=True: AddressSanitizer + UndefinedBehaviorSanitizer over the host planner / executor (ry_plan.cpp / ry_exec.cpp / ry_vc.cpp: plans, LRU graph slots, pointer
    arithmetic) and the kernels' index arithmetic -> tests/emu/libry355_emu_asan.so, run by scripts/asan_emu.sh."""
    if sanitize:
        suffix = suffix or '_asan'
    deps = SOURCES + [EMU_DIR / 'ry_emu.h', EMU_DIR / 'ry_emu.cpp']
    if defs and not suffix:
        raise ValueError('an experiment build (extra -D switches) needs a suffix: it must never replace the test emulator library')
    emu_lib = EMU_LIB if not suffix else EMU_LIB.with_name('libry355_emu%s.so' % suffix)
    if force or _stale(emu_lib, deps):
        cxx = '/opt/rocm/lib/llvm/bin/clang++'
        if not Path(cxx).exists():
            cxx = shutil.which('clang++') or shutil.which('g++')
        # plain -O2 on purpose: with AVX-512 enabled (-march=native on this host) ROCm's clang drops the tail of the fminf chain in
        # ry_pad_min_rows<16> (the remainder after the 8-wide gather is only run when a NaN was seen) -- found with the emulator tests
        san = ['-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-g', '-shared-libasan'] if sanitize else []
        objs = _compile_units(cxx, ['-x', 'c++', '-DRY_HOST_EMU', '-O1' if sanitize else '-O2', '-std=c++17', '-fPIC', '-pthread', '-Wno-psabi'] + san +
                              ['-D' + d for d in defs] + ['-I' + str(EMU_DIR), '-I' + str(CSRC)], ROOT / 'gpurun_out' / '_obj', '_emu' + suffix, [EMU_DIR / 'ry_emu.cpp'])
        tmp = emu_lib.with_suffix('.so.tmp%d' % os.getpid())              # several test workers may build at once: link aside, then rename
        subprocess.run([cxx, '-shared', '-fPIC', '-pthread'] + san + objs + ['-o', str(tmp)], check=True, cwd=str(ROOT))
        os.replace(str(tmp), str(emu_lib))
    return emu_lib


if __name__ == '__main__':
    print(build_product(force=bool(os.environ.get('FORCE'))))
