#!/bin/bash
# Round 5, GPU call A: the new output-stationary kernel on the hardware -- its GPU tests (block map of v_mfma_f32_4x4x1_16B_f32, operator
# cases, BASELINE layer sizes), then the slice sweep of the six bottom layers at the 300- and the 100-frame window.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "os_ or 4x4x1" > $O/pytest_os.txt 2>&1; echo "pytest os exit $?"; tail -15 $O/pytest_os.txt
timeout 900 python scripts/gpu_r5_os_sweep.py 300 $O/os_sweep_n300.txt > $O/sweep300.log 2>&1; echo "sweep 300 exit $?"; tail -40 $O/sweep300.log
timeout 600 python scripts/gpu_r5_os_sweep.py 100 $O/os_sweep_n100.txt > $O/sweep100.log 2>&1; echo "sweep 100 exit $?"; tail -12 $O/sweep100.log
