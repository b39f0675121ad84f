#!/bin/bash
# Round 5, GPU call B: what bounds ry_c2d_os -- the streaming calibration (tools/stream_probe: lockstep against rotated walks, private against
# shared regions) and the slice sweep with the rotated start of the K walk on and off.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_b; mkdir -p $O
for a in "256 160" "256 640" "128 128" "512 64"; do timeout 120 tools/stream_probe $a 10 >> $O/stream_probe.txt 2>&1; done; cat $O/stream_probe.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "os_ or 4x4x1" > $O/pytest_os.txt 2>&1; echo "pytest os exit $?"; tail -3 $O/pytest_os.txt
RY_OS2_ROT=1 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_rot1.txt > $O/sweep300_rot1.log 2>&1; echo "sweep 300 rot exit $?"; grep -A8 "^encoder/c7\|^decoder/c0\|^encoder/c6\|^decoder/c1\|^encoder/c5\|^decoder/c2" $O/sweep300_rot1.log | head -80; tail -22 $O/sweep300_rot1.log
RY_OS2_ROT=0 SWEEP_LAYERS=7,6 timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300_rot0.txt > $O/sweep300_rot0.log 2>&1; echo "sweep 300 norot exit $?"; grep -A5 "^encoder/c7\|^encoder/c6" $O/sweep300_rot0.log | head -20
RY_OS2_ROT=1 timeout 600 python scripts/gpu_r5_os_sweep.py 100 $O/os_sweep_n100.txt > $O/sweep100.log 2>&1; echo "sweep 100 exit $?"; tail -22 $O/sweep100.log
