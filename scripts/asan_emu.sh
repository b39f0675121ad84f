#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host executor and the kernels' index arithmetic (round-4 verdict):
# the emulator build of the same sources with -fsanitize=address,undefined, run under the emulator test files.  CPU only, ~10-20 min.
#   scripts/asan_emu.sh [pytest args]      default: tests/test_kernels_emu.py tests/test_vc_api.py
cd "$(dirname "$0")/.."
LIB=$(python -c "from realtime_yukarin_amd import build; print(build.build_emu(sanitize=True))" 2>/dev/null | tail -1) || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
[ $# -eq 0 ] && set -- tests/test_kernels_emu.py tests/test_vc_api.py
# the emulator's fibers run on malloc'ed stacks with a hand-written context switch: no fake stacks (detect_stack_use_after_return=0); leaks of the
# Python interpreter itself are not ours (detect_leaks=0)
LD_PRELOAD=$RT RY_EMU_LIB=$LIB ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 python -m pytest "$@" -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/asan_emu.log | tail -40
