"""Build-time check on the gfx950 machine code of the product library (CPU: llvm-objdump of the code object, nothing runs).

ry_c2d_os's LDS-DMA pixel path (XL) reads LDS bytes that the wave's OWN earlier global_load_lds wrote and times that read with a hand-written
`s_waitcnt vmcnt(after * (MT4 + NT4))` (ry_kernels.h, consume()): the count is right only while the compiler emits exactly MT4 + NT4 vector-memory
loads per K unit, in program order, between two such waits -- a merged or hoisted load would let the wait pass before the slot has landed, and
neither the emulator nor a CPU test can see that (round 5: hipcc 7.2's own timing read a slot early on the MI355X only, profiles/r05/n_xl.txt;
round-5 advisor).  This test counts them in the disassembly of every XL instantiation."""
import re
import subprocess
from pathlib import Path

import pytest

from realtime_yukarin_amd import build

LLVM = Path('/opt/rocm/lib/llvm/bin')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


@pytest.fixture(scope='module')
def code_objects(tmp_path_factory):
    """the gfx950 code objects inside libry355.so (.hip_fatbin holds one offload bundle per translation unit)"""
    if not (LLVM / 'llvm-objdump').exists():
        pytest.skip('no llvm-objdump in this image')
    d = tmp_path_factory.mktemp('co')
    fat = d / 'fat.bin'
    subprocess.run([str(LLVM / 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', str(build.build_product()), str(fat)], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, 'no offload bundle in .hip_fatbin'
    out = []
    for i, s in enumerate(starts):
        piece = d / ('bundle%d.bin' % i)
        piece.write_bytes(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        elf = d / ('unit%d.elf' % i)
        subprocess.run([str(LLVM / 'clang-offload-bundler'), '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + str(piece), '--output=' + str(elf)], check=True)
        out.append(elf)
    return out


def kernels(elf, pattern):
    """{symbol: [instruction text, ...]} of the functions whose mangled name matches `pattern`"""
    text = subprocess.run([str(LLVM / 'llvm-objdump'), '-d', '--no-show-raw-insn', str(elf)], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = m.group(1) if re.search(pattern, m.group(1)) else None
            if cur:
                out[cur] = []
            continue
        if cur and line.strip():
            out[cur].append(line.split('//')[0].strip())
    return out


def test_xl_waits_see_exactly_one_unit_of_loads(code_objects):
    found = {}
    for elf in code_objects:
        found.update(kernels(elf, r'^_Z9ry_c2d_osILi\d+ELi\d+ELi\d+ELi\d+ELb1EE'))
    assert len(found) >= 10, sorted(found)                   # every (MT4, NT4, WAVES, 2) slice whose ring fits has an XL instantiation
    for name, ins in found.items():
        mt4, nt4 = (int(v) for v in re.match(r'^_Z9ry_c2d_osILi(\d+)ELi(\d+)E', name).groups())
        unit = mt4 + nt4
        checked, loads, clean, have = 0, 0, False, False
        n_dma = sum(1 for i in ins if i.startswith('global_load_lds_dwordx4'))
        n_flt = sum(1 for i in ins if re.match(r'global_load_dwordx4\b', i))
        assert n_dma % mt4 == 0 and n_flt % nt4 == 0 and n_dma // mt4 == n_flt // nt4, (name, n_dma, n_flt)     # whole units only: nothing merged, nothing dropped
        for i in ins:
            if re.match(r'global_load_(lds_)?dword', i):
                loads += 1
            elif re.match(r's_(c)?branch', i) or i.startswith('s_setpc'):
                clean = False                                # another path joins or leaves between the two waits: not a straight-line pair
            elif re.fullmatch(r's_waitcnt vmcnt\(%d\)' % unit, i):
                if have and clean:
                    assert loads == unit, '%s: %d vector-memory loads between two s_waitcnt vmcnt(%d), the wait assumes %d' % (name, loads, unit, unit)
                    checked += 1
                have, clean, loads = True, True, 0
        assert checked >= 1, '%s: no straight-line pair of s_waitcnt vmcnt(%d) found -- the K loop no longer has the shape this check knows' % (name, unit)
