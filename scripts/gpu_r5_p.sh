#!/bin/bash
# Round 5, GPU call P: the LDS-DMA pixel path with the ring refilled in place (two slots, explicit lgkmcnt(0) between the reads of a slot and the DMA
# over it): every slice against the implicit GEMM, then planner defaults with and without the path, twice each, interleaved.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "os_ or 4x4x1 or stage2 or config" 2>&1 | tail -1
for r in 1 2; do
RY_OS2_XL=0 SWEEP_LAYERS=none timeout 600 python scripts/gpu_r5_os_sweep.py 300 $O/os_defaults_n300_xl0_$r.txt > $O/sweep300_xl0_$r.log 2>&1; tail -9 $O/sweep300_xl0_$r.log | grep -v "encoder/c5\|decoder/c2"
RY_OS2_XL=1 SWEEP_LAYERS=none timeout 600 python scripts/gpu_r5_os_sweep.py 300 $O/os_defaults_n300_xl1_$r.txt > $O/sweep300_xl1_$r.log 2>&1; tail -9 $O/sweep300_xl1_$r.log | grep -v "encoder/c5\|decoder/c2"
done
