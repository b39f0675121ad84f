#!/bin/bash
# Round 5, GPU call E: ry_c2d_os with the round-structured K loop (rounds of four units, immediate offsets, 4 / 8 / 16 waves per workgroup):
# GPU tests, slice sweeps at 300 and 100 frames, ablations on the diagnostic build.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "os_ or 4x4x1" > $O/pytest_os.txt 2>&1; echo "pytest os exit $?"; tail -3 $O/pytest_os.txt
timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300.txt > $O/sweep300.log 2>&1; echo "sweep 300 exit $?"; grep -A7 "^encoder/c7\|^decoder/c0\|^encoder/c6\|^decoder/c1\|^encoder/c5\|^decoder/c2" $O/sweep300.log | head -60; tail -22 $O/sweep300.log
timeout 600 python scripts/gpu_r5_os_sweep.py 100 $O/os_sweep_n100.txt > $O/sweep100.log 2>&1; echo "sweep 100 exit $?"; tail -22 $O/sweep100.log
timeout 600 python scripts/gpu_r5_os_ablate.py 300 $O/os_ablate_n300.txt 2>&1 | tail -16
