#!/bin/bash
# round 6: the ablation builds of the Winograd K loop, one process each (libry355_abl<bits>.so built here: build.build_product(defs=['RY_WINO_ABL=<bits>'], suffix='_abl<bits>'))
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_abl; mkdir -p $O
echo "# us of the Winograd launch (layer grid us) under RY_WINO=12:1:2:1,13:1:2:1,14:1:2:1,1:1:2:1,3:1:2:1 | 12:1:2:2,13:1:2:3,14:2:2:1,1:2:1:1,3:1:2:5; bits: 1 no transform, 2 no patch reads, 4 no filter reads, 8 no barrier, 32 no MFMAs" > $O/abl.txt
for b in 0 1 3 4 7 8 15 32 0; do timeout 300 python scripts/gpu_r6_abl.py $b 300 2>$O/err_$b.txt | grep "^abl" >> $O/abl.txt; done
cat $O/abl.txt
