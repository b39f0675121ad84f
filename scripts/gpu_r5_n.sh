#!/bin/bash
# Round 5, GPU call N: the LDS-DMA pixel path of ry_c2d_os with one LDS slot per unit of a round (a slot is refilled two units after it was
# read): every slice six times against the implicit GEMM, the GPU parity tests of the kernel, the slice sweep at 300 frames.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_n; mkdir -p $O
RY_OS2_XL=1 timeout 300 python scripts/_xl_tmp.py 2>&1 | tail -2
RY_OS2_XL=0 timeout 300 python scripts/_xl_tmp.py 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "os_ or 4x4x1 or stage2 or config" 2>&1 | tail -1
RY_OS2_XL=1 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_xl1.txt > $O/sweep300_xl1.log 2>&1; echo "sweep xl1 exit $?"; grep -A5 "^encoder/c7\|^decoder/c0\|^encoder/c6\|^decoder/c1\|^encoder/c5\|^decoder/c2" $O/sweep300_xl1.log | head -50; tail -12 $O/sweep300_xl1.log
