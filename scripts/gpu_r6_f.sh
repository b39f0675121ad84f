#!/bin/bash
# round 6, call f: partial vmcnt waits (the next patch's DMA stays in flight across a barrier): parity first, then the defaults and a few plans in turn
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wino or stage2_syn64_convert" > $O/pytest_wino.txt 2>&1; echo "pytest wino exit $?"; tail -3 $O/pytest_wino.txt
timeout 1500 python scripts/gpu_r6_plan_ab.py 300 3 $O/plan_ab_n300.txt - "env:RY_WINO=12:1:2:3" "env:RY_WINO=13:1:2:3" "env:RY_WINO=3:2:4:5" "env:RY_WINOGRAD=0" > $O/ab.log 2>&1; echo "ab exit $?"; grep "^#" $O/plan_ab_n300.txt | cut -c1-150
