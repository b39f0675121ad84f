#!/bin/bash
# round 6, call i: the two-blocks-per-wave shapes of ry_wino_ldsdma (TN = 2: cfg 3 / 4, one wave per SIMD, raw fragments prefetched): parity, then the plan sweep at 300 frames
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wino" > $O/pytest_wino.txt 2>&1; echo "pytest wino exit $?"; tail -5 $O/pytest_wino.txt
SWEEP_CFGS=3,4 timeout 2400 python scripts/gpu_r6_wino.py 300 $O/wino_sweep_n300.txt 10 > $O/sweep.log 2>&1; echo "sweep exit $?"; grep "^#" $O/wino_sweep_n300.txt | tail -60
