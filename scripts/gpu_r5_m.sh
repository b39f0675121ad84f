#!/bin/bash
# Round 5, GPU call M: ry_c2d_os with the pixels through the LDS (RY_OS2_XL=1, compiler-timed: vmcnt(0) in front of every unit's reads) against
# straight into registers (=0): slice sweep at 300 frames.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "os_ or 4x4x1" 2>&1 | tail -1
RY_OS2_XL=1 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_xl1.txt > $O/sweep300_xl1.log 2>&1; echo "sweep xl1 exit $?"; grep -A5 "^encoder/c7\|^decoder/c0\|^encoder/c6\|^decoder/c1\|^encoder/c5\|^decoder/c2" $O/sweep300_xl1.log | head -50; tail -12 $O/sweep300_xl1.log
RY_OS2_XL=0 SWEEP_LAYERS=6,9,5 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_xl0.txt > $O/sweep300_xl0.log 2>&1; grep -A3 "^encoder/c6\|^decoder/c1\|^encoder/c5" $O/sweep300_xl0.log | head -16
