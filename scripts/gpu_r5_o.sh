#!/bin/bash
# Round 5, GPU call O: the whole GPU suite with the LDS-DMA pixel path of ry_c2d_os (explicit waits) as the default, the planner's picks at 100 frames,
# the default bench twice.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_o; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gpu.txt
SWEEP_LAYERS=none timeout 600 python scripts/gpu_r5_os_sweep.py 100 $O/os_defaults_n100.txt > $O/sweep100.log 2>&1; tail -9 $O/sweep100.log
RY_OS2_XL=0 SWEEP_LAYERS=none timeout 600 python scripts/gpu_r5_os_sweep.py 300 $O/os_defaults_n300_xl0.txt > $O/sweep300_xl0.log 2>&1; tail -9 $O/sweep300_xl0.log
SWEEP_LAYERS=none timeout 600 python scripts/gpu_r5_os_sweep.py 300 $O/os_defaults_n300_xl1.txt > $O/sweep300_xl1.log 2>&1; tail -9 $O/sweep300_xl1.log
timeout 600 python bench.py > $O/bench1.json 2> $O/bench1.err; cat $O/bench1.json
timeout 600 python bench.py > $O/bench2.json 2> $O/bench2.err; cat $O/bench2.json
