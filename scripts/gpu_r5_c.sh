#!/bin/bash
# Round 5, GPU call C: non-temporal filter streams -- ry_c2d_os with RY_OS2_NT=1 (full slice sweep) against 0, the implicit GEMM's gather
# variant with non-temporal filter tiles (RY_IGEMM_BNT) on the bottom layers.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_c; mkdir -p $O
RY_OS2_NT=1 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_nt1.txt > $O/sweep300_nt1.log 2>&1; echo "sweep 300 nt exit $?"; head -24 $O/sweep300_nt1.log; grep -A8 "^encoder/c7\|^decoder/c0\|^encoder/c6\|^decoder/c1\|^encoder/c5\|^decoder/c2" $O/sweep300_nt1.log | head -80; tail -22 $O/sweep300_nt1.log
RY_OS2_NT=0 SWEEP_LAYERS=7,6,9 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_nt0.txt > $O/sweep300_nt0.log 2>&1; echo "sweep 300 nt0 exit $?"; grep -A4 "^encoder/c7\|^encoder/c6\|^decoder/c1" $O/sweep300_nt0.log | head -20
RY_OS2_NT=1 timeout 600 python scripts/gpu_r5_os_sweep.py 100 $O/os_sweep_n100.txt > $O/sweep100.log 2>&1; echo "sweep 100 exit $?"; tail -22 $O/sweep100.log
