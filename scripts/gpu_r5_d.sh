#!/bin/bash
# Round 5, GPU call D: ablations of ry_c2d_os (RY_OS2_DBG) on the six bottom layers.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_d; mkdir -p $O
timeout 600 python scripts/gpu_r5_os_ablate.py 300 $O/os_ablate_n300.txt 2>&1 | tail -20
ABLATE_CFG='6:3:2:8:2,7:1:2:8:2,8:3:2:8:2,9:3:4:4:4' timeout 600 python scripts/gpu_r5_os_ablate.py 300 $O/os_ablate_n300_b.txt 2>&1 | tail -16
